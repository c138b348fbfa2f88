"""Pack a diffusers-style state dict (the keys the reference loads,
lakonlab/pipelines/arcflow_loader.py:241-263 and SURVEY App. B) into the engine's fused weight set.

Fusions (all done once at load time, on the device):
  * to_k | to_v | to_q (and add_k|add_v|add_q) stacked on the output dim -> one QKV GEMM per stream;
    the FLUX single block additionally stacks proj_mlp: rows k|v|q|mlp.  (q last: attention writes O
    over Q so that [O | mlp] is the contiguous-K operand of proj_out.)
  * every AdaLN modulation linear of the network stacked into one [n_mod, D] matrix.
  * the three ArcFlow heads stacked into one [K*C + K*L + (K-1)*L (+pad), D] matrix.
  * LoRA (lora_A/lora_B, scale alpha/r = 1: arcflux.py:295-301) is merged into the base weight in
    fp32 and rounded once to bf16 (the reference keeps the side GEMMs un-fused; merge changes only
    rounding order -- see DESIGN.md).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

Tensor = torch.Tensor


def merge_lora(sd: Dict[str, Tensor], lora: Optional[Dict[str, Tensor]], scale: float = 1.0) -> Dict[str, Tensor]:
    """Return a copy of ``sd`` with every ``<mod>.lora_A.weight``/``lora_B.weight`` pair folded in."""
    if not lora:
        return sd
    out = dict(sd)
    mods = sorted({k.rsplit('.lora_', 1)[0] for k in lora if '.lora_A' in k})
    for m in mods:
        a = lora.get(m + '.lora_A.weight', lora.get(m + '.lora_A.default.weight'))
        b = lora.get(m + '.lora_B.weight', lora.get(m + '.lora_B.default.weight'))
        key = m + '.weight'
        if key not in out:
            key = m + '.base_layer.weight'
        if a is None or b is None or key not in out:
            raise KeyError(f'LoRA module {m}: missing lora_A/lora_B or base weight')
        w = out[key]
        out[m + '.weight'] = _merge_one(w, a, b, scale)
    return out


def _merge_one(w: Tensor, a: Tensor, b: Tensor, scale: float) -> Tensor:
    """W + scale * B A, the sum accumulated in fp32 and rounded once to W's dtype.  On the GPU this is the library's own
    fp32-accumulating GEMM (`afx_linear_bf16_f32out`, C += B . (A^T)^T with the rank as the contraction, zero-padded to the
    kernel's K granule): its OPERANDS are bf16, so `B * scale` and `A` are rounded to bf16 first -- exact for the released bf16
    adapters, a 2^-9 relative perturbation of the delta B A for fp32 / fp16 adapter files (the same rounding the training engine
    applies to its bf16 working copies of A and B; bounded in tests/test_pipeline.py::test_lora_merge_gpu_vs_host_branch).  The host
    branch forms the delta from the fp32 tensors; it only serves weights packed on a machine without a device (the CPU tests)."""
    if not (w.is_cuda or (torch.cuda.is_available() and a.is_cuda)):
        delta = (b.to(torch.float32) @ a.to(torch.float32)) * scale
        return (w.float() + delta).to(w.dtype)
    from . import ops
    dev = w.device if w.is_cuda else a.device
    r = a.shape[0]
    rp = (r + 63) // 64 * 64
    bb = torch.zeros(b.shape[0], rp, dtype=torch.bfloat16, device=dev)
    bb[:, :r] = (b.to(dev, torch.float32) * scale).to(torch.bfloat16)
    at = torch.zeros(a.shape[1], rp, dtype=torch.bfloat16, device=dev)
    at[:, :r] = a.to(dev, torch.bfloat16).t()
    acc = w.to(dev, torch.float32).contiguous()
    ops.linear_f32out(bb, at, out=acc, accumulate=True)
    return acc.to(w.dtype).to(w.device)


def _bf16(t: Tensor, device) -> Tensor:
    return t.to(device=device, dtype=torch.bfloat16).contiguous()


def _cat(sd, names, device, dim=0) -> Tensor:
    return torch.cat([sd[n].to(device=device, dtype=torch.bfloat16) for n in names], dim=dim).contiguous()


def pack_head(sd, device, K: int, C: int, L: int, teacher: bool = False):
    if teacher:
        return _bf16(sd['proj_out.weight'], device), _bf16(sd['proj_out.bias'], device)
    w = _cat(sd, ['proj_out_means.weight', 'proj_out_logweights.weight', 'proj_out_loggamma.weight'], device)
    b = _cat(sd, ['proj_out_means.bias', 'proj_out_logweights.bias', 'proj_out_loggamma.bias'], device)
    n = w.shape[0]
    npad = (n + 7) // 8 * 8
    if npad != n:
        w = torch.cat([w, w.new_zeros(npad - n, w.shape[1])]).contiguous()
        b = torch.cat([b, b.new_zeros(npad - n)]).contiguous()
    return w, b


def pack_flux(sd: Dict[str, Tensor], num_double: int, num_single: int, device, K=16, C=64, L=4,
              guidance: bool = True, teacher: bool = False) -> Dict[str, Tensor]:
    p: Dict[str, Tensor] = {}

    def lin(dst, src):
        p[dst + '.weight'] = _bf16(sd[src + '.weight'], device)
        p[dst + '.bias'] = _bf16(sd[src + '.bias'], device)
    lin('x_in', 'x_embedder')
    lin('ctx_in', 'context_embedder')
    for tag, nm in (('t', 'timestep_embedder'), ('p', 'text_embedder')) + ((('g', 'guidance_embedder'),) if guidance else ()):
        lin(f'temb.{tag}.l1', f'time_text_embed.{nm}.linear_1')
        lin(f'temb.{tag}.l2', f'time_text_embed.{nm}.linear_2')
    mod_w, mod_b = [], []
    for i in range(num_double):
        s, d = f'transformer_blocks.{i}.', f'd{i}.'
        for nm in ('norm1.linear', 'norm1_context.linear'):
            mod_w.append(s + nm + '.weight'); mod_b.append(s + nm + '.bias')
        for stream, (k, v, q, o, ff) in (('img', ('attn.to_k', 'attn.to_v', 'attn.to_q', 'attn.to_out.0', 'ff')),
                                         ('txt', ('attn.add_k_proj', 'attn.add_v_proj', 'attn.add_q_proj',
                                                  'attn.to_add_out', 'ff_context'))):
            p[d + stream + '_qkv.weight'] = _cat(sd, [s + k + '.weight', s + v + '.weight', s + q + '.weight'], device)
            p[d + stream + '_qkv.bias'] = _cat(sd, [s + k + '.bias', s + v + '.bias', s + q + '.bias'], device)
            lin(d + stream + '_out', s + o)
            lin(d + stream + '_mlp1', s + ff + '.net.0.proj')
            lin(d + stream + '_mlp2', s + ff + '.net.2')
        p[d + 'qknorm'] = torch.stack([sd[s + f'attn.{n}.weight'].float() for n in
                                       ('norm_q', 'norm_k', 'norm_added_q', 'norm_added_k')]).to(device).contiguous()
    for i in range(num_single):
        s, d = f'single_transformer_blocks.{i}.', f's{i}.'
        mod_w.append(s + 'norm.linear.weight'); mod_b.append(s + 'norm.linear.bias')
        names = ['attn.to_k', 'attn.to_v', 'attn.to_q', 'proj_mlp']
        p[d + 'fused.weight'] = _cat(sd, [s + n + '.weight' for n in names], device)
        p[d + 'fused.bias'] = _cat(sd, [s + n + '.bias' for n in names], device)
        lin(d + 'out', s + 'proj_out')
        p[d + 'qknorm'] = torch.stack([sd[s + 'attn.norm_q.weight'].float(),
                                       sd[s + 'attn.norm_k.weight'].float()]).to(device).contiguous()
    mod_w.append('norm_out.linear.weight'); mod_b.append('norm_out.linear.bias')
    p['mod.weight'] = _cat(sd, mod_w, device)
    p['mod.bias'] = _cat(sd, mod_b, device)
    p['head.weight'], p['head.bias'] = pack_head(sd, device, K, C, L, teacher)
    return p


def pack_qwen(sd: Dict[str, Tensor], num_layers: int, device, K=16, C=64, L=4, teacher: bool = False
              ) -> Dict[str, Tensor]:
    p: Dict[str, Tensor] = {}

    def lin(dst, src):
        p[dst + '.weight'] = _bf16(sd[src + '.weight'], device)
        p[dst + '.bias'] = _bf16(sd[src + '.bias'], device)
    lin('x_in', 'img_in')
    lin('ctx_in', 'txt_in')
    p['txt_norm.weight'] = sd['txt_norm.weight'].to(device=device, dtype=torch.float32).contiguous()
    lin('temb.t.l1', 'time_text_embed.timestep_embedder.linear_1')
    lin('temb.t.l2', 'time_text_embed.timestep_embedder.linear_2')
    mod_w, mod_b = [], []
    for i in range(num_layers):
        s, d = f'transformer_blocks.{i}.', f'd{i}.'
        for nm in ('img_mod.1', 'txt_mod.1'):
            mod_w.append(s + nm + '.weight'); mod_b.append(s + nm + '.bias')
        for stream, (k, v, q, o, ff) in (('img', ('attn.to_k', 'attn.to_v', 'attn.to_q', 'attn.to_out.0', 'img_mlp')),
                                         ('txt', ('attn.add_k_proj', 'attn.add_v_proj', 'attn.add_q_proj',
                                                  'attn.to_add_out', 'txt_mlp'))):
            p[d + stream + '_qkv.weight'] = _cat(sd, [s + k + '.weight', s + v + '.weight', s + q + '.weight'], device)
            p[d + stream + '_qkv.bias'] = _cat(sd, [s + k + '.bias', s + v + '.bias', s + q + '.bias'], device)
            lin(d + stream + '_out', s + o)
            lin(d + stream + '_mlp1', s + ff + '.net.0.proj')
            lin(d + stream + '_mlp2', s + ff + '.net.2')
        p[d + 'qknorm'] = torch.stack([sd[s + f'attn.{n}.weight'].float() for n in
                                       ('norm_q', 'norm_k', 'norm_added_q', 'norm_added_k')]).to(device).contiguous()
    mod_w.append('norm_out.linear.weight'); mod_b.append('norm_out.linear.bias')
    p['mod.weight'] = _cat(sd, mod_w, device)
    p['mod.bias'] = _cat(sd, mod_b, device)
    p['head.weight'], p['head.bias'] = pack_head(sd, device, K, C, L, teacher)
    return p


# ------------------------------------------------------------------------------------------------
# Synthetic weights generated directly in the packed layout on the GPU (bench / scaling runs: there is
# no network for checkpoints, and 12-20 B parameters are too slow to draw on the host).
def random_packed(family: str, num_double: int, num_single: int, device, heads: int = 24, in_channels: int = 64,
                  joint_dim: int = 4096, pooled_dim: int = 768, guidance: bool = True, K: int = 16, L: int = 4,
                  seed: int = 0, teacher: bool = False) -> Dict[str, Tensor]:
    """Random-init weights of the FLUX / Qwen-Image architecture, N(0, 0.02^2) linears, unit-ish norms."""
    D = heads * 128
    g = torch.Generator(device=device).manual_seed(seed)
    p: Dict[str, Tensor] = {}

    def lin(name, out_f, in_f, std=0.02, bstd=0.02):
        w = torch.empty(out_f, in_f, dtype=torch.bfloat16, device=device)
        # draw in slabs: a [12288, 15360] fp32 temporary would cost 750 MB
        step = max(1, (1 << 26) // in_f)
        for r0 in range(0, out_f, step):
            r1 = min(out_f, r0 + step)
            w[r0:r1] = (torch.randn(r1 - r0, in_f, generator=g, device=device) * std).to(torch.bfloat16)
        p[name + '.weight'] = w
        p[name + '.bias'] = (torch.randn(out_f, generator=g, device=device) * bstd).to(torch.bfloat16)

    def norms(name, n):
        p[name] = (1 + 0.02 * torch.randn(n, 128, generator=g, device=device)).float().contiguous()
    lin('x_in', D, in_channels, std=0.1)
    lin('ctx_in', D, joint_dim)
    if family == 'qwen':
        p['txt_norm.weight'] = (1 + 0.02 * torch.randn(joint_dim, generator=g, device=device)).float()
    lin('temb.t.l1', D, 256, std=0.05); lin('temb.t.l2', D, D, std=0.03)
    if family == 'flux':
        if guidance:
            lin('temb.g.l1', D, 256, std=0.05); lin('temb.g.l2', D, D, std=0.03)
        lin('temb.p.l1', D, pooled_dim, std=0.05); lin('temb.p.l2', D, D, std=0.03)
    n_mod = num_double * 12 * D + num_single * 3 * D + 2 * D
    lin('mod', n_mod, D, bstd=0.3)
    for i in range(num_double):
        for s in ('img', 'txt'):
            lin(f'd{i}.{s}_qkv', 3 * D, D); lin(f'd{i}.{s}_out', D, D)
            lin(f'd{i}.{s}_mlp1', 4 * D, D); lin(f'd{i}.{s}_mlp2', D, 4 * D)
        norms(f'd{i}.qknorm', 4)
    for i in range(num_single):
        lin(f's{i}.fused', 7 * D, D); lin(f's{i}.out', D, 5 * D)
        norms(f's{i}.qknorm', 2)
    raw = in_channels if teacher else K * in_channels + K * L + (K - 1) * L
    lin('head', (raw + 7) // 8 * 8, D)
    return p


def init_arcflow_heads_from_teacher(sd: Dict[str, Tensor], K: int = 16, L: int = 4, noise_std: float = 0.05,
                                    generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    """Student head initialisation from the teacher's ``proj_out`` (reference
    lakonlab/models/architecture/arcflow/arcflux.py:328-341 and :103-132): means = K tiled copies of proj_out with a
    per-(component, channel) bias jitter shared by the p^2 sub-pixels, log-weights zero, log-gamma weight zero with
    the log-spaced rate bias.  Returns a copy of ``sd`` with the three ``proj_out_*`` linears added."""
    import math
    out = dict(sd)
    w, b = sd['proj_out.weight'], sd['proj_out.bias']
    C, D = w.shape
    out['proj_out_means.weight'] = w[None].expand(K, -1, -1).reshape(K * C, D).clone()
    jitter = torch.randn(K * C // L, generator=generator) * noise_std
    out['proj_out_means.bias'] = (b[None].expand(K, -1).reshape(K * C).float()
                                  + jitter[:, None].expand(-1, L).flatten()).to(b.dtype)
    out['proj_out_logweights.weight'] = torch.zeros(K * L, D, dtype=w.dtype)
    out['proj_out_logweights.bias'] = torch.zeros(K * L, dtype=b.dtype)
    out['proj_out_loggamma.weight'] = torch.zeros((K - 1) * L, D, dtype=w.dtype)
    rates = torch.logspace(math.log10(0.2), math.log10(4.0), K - 1, base=10)
    out['proj_out_loggamma.bias'] = torch.log(rates).unsqueeze(1).repeat(1, L).flatten().to(b.dtype)
    return out


# ------------------------------------------------------------------------------------------------
# Expected diffusers state-dict layout (names -> shapes) of the transformers this engine ingests: what
# ``tools/check_snapshot.py`` validates a real checkpoint's safetensors headers against (SURVEY 8c self-check 1: weight names and
# shapes of FLUX.1-dev / Qwen-Image as the reference's ``from_pretrained`` + ``load_arcflow_adapter`` consume them,
# lakonlab/pipelines/arcflow_loader.py:241-263).
def expected_transformer_keys(family: str, cfg: Dict, student: bool = False, K: int = 16, L: int = 4) -> Dict[str, tuple]:
    """family 'flux' | 'qwen'; cfg = the transformer's config.json (diffusers field names).  student: ArcFlow heads
    (proj_out_means / _logweights / _loggamma) instead of the plain ``proj_out``."""
    H, hd = cfg.get('num_attention_heads', 24), cfg.get('attention_head_dim', 128)
    D = H * hd
    C = cfg.get('in_channels', 64)
    J = cfg.get('joint_attention_dim', 4096 if family == 'flux' else 3584)
    e: Dict[str, tuple] = {}

    def lin(name, o, i):
        e[name + '.weight'] = (o, i)
        e[name + '.bias'] = (o,)
    if family == 'flux':
        lin('x_embedder', D, C)
        lin('context_embedder', D, J)
        names = ['timestep_embedder', 'text_embedder'] + (['guidance_embedder'] if cfg.get('guidance_embeds', True) else [])
        for nm in names:
            lin(f'time_text_embed.{nm}.linear_1', D, cfg.get('pooled_projection_dim', 768) if nm == 'text_embedder' else 256)
            lin(f'time_text_embed.{nm}.linear_2', D, D)
        for i in range(cfg.get('num_layers', 19)):
            p = f'transformer_blocks.{i}.'
            lin(p + 'norm1.linear', 6 * D, D)
            lin(p + 'norm1_context.linear', 6 * D, D)
            for nm in ('to_q', 'to_k', 'to_v', 'add_q_proj', 'add_k_proj', 'add_v_proj', 'to_out.0', 'to_add_out'):
                lin(p + 'attn.' + nm, D, D)
            for nm in ('norm_q', 'norm_k', 'norm_added_q', 'norm_added_k'):
                e[p + f'attn.{nm}.weight'] = (hd,)
            for ff in ('ff', 'ff_context'):
                lin(p + ff + '.net.0.proj', 4 * D, D)
                lin(p + ff + '.net.2', D, 4 * D)
        for i in range(cfg.get('num_single_layers', 38)):
            p = f'single_transformer_blocks.{i}.'
            lin(p + 'norm.linear', 3 * D, D)
            for nm in ('to_q', 'to_k', 'to_v'):
                lin(p + 'attn.' + nm, D, D)
            e[p + 'attn.norm_q.weight'] = (hd,)
            e[p + 'attn.norm_k.weight'] = (hd,)
            lin(p + 'proj_mlp', 4 * D, D)
            lin(p + 'proj_out', D, 5 * D)
    elif family == 'qwen':
        lin('img_in', D, C)
        lin('txt_in', D, J)
        e['txt_norm.weight'] = (J,)
        lin('time_text_embed.timestep_embedder.linear_1', D, 256)
        lin('time_text_embed.timestep_embedder.linear_2', D, D)
        for i in range(cfg.get('num_layers', 60)):
            p = f'transformer_blocks.{i}.'
            lin(p + 'img_mod.1', 6 * D, D)
            lin(p + 'txt_mod.1', 6 * D, D)
            for nm in ('to_q', 'to_k', 'to_v', 'add_q_proj', 'add_k_proj', 'add_v_proj', 'to_out.0', 'to_add_out'):
                lin(p + 'attn.' + nm, D, D)
            for nm in ('norm_q', 'norm_k', 'norm_added_q', 'norm_added_k'):
                e[p + f'attn.{nm}.weight'] = (hd,)
            for ff in ('img_mlp', 'txt_mlp'):
                lin(p + ff + '.net.0.proj', 4 * D, D)
                lin(p + ff + '.net.2', D, 4 * D)
    else:
        raise ValueError(family)
    lin('norm_out.linear', 2 * D, D)
    if student:
        lin('proj_out_means', K * C, D)
        lin('proj_out_logweights', K * L, D)
        lin('proj_out_loggamma', (K - 1) * L, D)
    else:
        lin('proj_out', C, D)
    return e


def check_state_shapes(shapes: Dict[str, tuple], expected: Dict[str, tuple]):
    """-> (missing names, [(name, got, want)] shape mismatches, unexpected names)."""
    missing = sorted(k for k in expected if k not in shapes)
    wrong = sorted((k, tuple(shapes[k]), tuple(expected[k])) for k in expected if k in shapes and tuple(shapes[k]) != tuple(expected[k]))
    extra = sorted(k for k in shapes if k not in expected)
    return missing, wrong, extra
