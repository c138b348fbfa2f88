"""Read the reference's training configs without mmcv.

The files under ``configs/*/arc*_2nfe_k16.py`` are plain Python: module-level names are the config keys, ``_base_``
lists files merged underneath (dict values merge recursively, everything else is replaced -- mmcv ``Config``
semantics, ``_delete_=True`` drops the inherited dict).  ``distill_setup`` maps the keys this engine consumes
(configs/flux/arcflux_2nfe_k16.py:13-149, configs/flux/_ddp_train.py:13-36) onto ``DistillConfig`` + engine kwargs.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Tuple

from .distill import DistillConfig

_FAMILY = {'ArcFluxTransformer2DModel': 'flux', 'ArcQwenImageTransformer2DModel': 'qwen'}


def _merge(base: Dict[str, Any], over: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        elif isinstance(v, dict):
            out[k] = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
        else:
            out[k] = v
    return out


def load_config(path: str) -> Dict[str, Any]:
    path = os.path.abspath(path)
    ns: Dict[str, Any] = {'__file__': path}
    with open(path, encoding='utf-8') as f:
        exec(compile(f.read(), path, 'exec'), ns)           # a config file is code the operator chose to run
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v) and not hasattr(v, '__loader__')}
    bases = cfg.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged: Dict[str, Any] = {}
    for b in bases:
        merged = _merge(merged, load_config(os.path.join(os.path.dirname(path), b)))
    return _merge(merged, cfg)


def apply_options(cfg: Dict[str, Any], options: Dict[str, Any]) -> Dict[str, Any]:
    """``--cfg-options a.b.c=v`` overrides (train.py:82-87 of the reference)."""
    out = cfg
    for dotted, v in options.items():
        keys = dotted.split('.')
        patch: Dict[str, Any] = {}
        cur = patch
        for k in keys[:-1]:
            cur[k] = {}
            cur = cur[k]
        cur[keys[-1]] = v
        out = _merge(out, patch)
    return out


def distill_setup(cfg: Dict[str, Any]) -> Tuple[str, Dict[str, Any], DistillConfig, Dict[str, Any]]:
    """-> (family, MMDiTEngine kwargs, DistillConfig, run parameters)."""
    diff = cfg['model']['diffusion']
    den = diff['denoising']
    if den['type'] not in _FAMILY:
        raise ValueError(f"unsupported denoising type {den['type']!r}")
    family = _FAMILY[den['type']]
    eng = dict(num_double=den['num_layers'], heads=den['num_attention_heads'], head_dim=den['attention_head_dim'],
               in_channels=den['in_channels'], joint_dim=den['joint_attention_dim'],
               num_gaussians=den['num_gaussians'], logweights_channels=den['logweights_channels'])
    if family == 'flux':
        eng.update(num_single=den['num_single_layers'], pooled_dim=den['pooled_projection_dim'],
                   guidance_embeds=den.get('guidance_embeds', True))
    if 'axes_dims_rope' in den:
        eng['axes_dims'] = tuple(den['axes_dims_rope'])
    tc = cfg.get('train_cfg', {})
    opt = cfg.get('optimizer', {}).get('diffusion', {})
    lr_cfg = cfg.get('lr_config', {})
    loss = diff.get('flow_loss', {})
    sampler = diff.get('timestep_sampler', {})
    ema = next((h for h in cfg.get('custom_hooks', []) if h.get('type') == 'ExponentialMovingAverageHookMod'), {})
    mults = opt.get('paramwise_cfg', {}).get('custom_keys', {})
    dc = DistillConfig(
        nfe=tc.get('nfe', 2), timestep_ratio=tc.get('timestep_ratio', 1.0), total_substeps=tc.get('total_substeps', 128),
        window_substeps=tc.get('window_substeps', 3), gm_dropout=tc.get('gm_dropout', 0.0),
        num_intermediate_states=tc.get('num_intermediate_states', 4), num_decay_iters=tc.get('num_decay_iters', 2000),
        shift=sampler.get('shift', 3.2), loss_scale=loss.get('rescale_cfg', {}).get('scale', 1.0),
        guidance=tc.get('distilled_guidance_scale', 3.5), teacher_guidance=tc.get('teacher_distilled_guidance_scale'),
        teacher_guidance_scale=tc.get('teacher_guidance_scale', 1.0),
        lr=opt.get('lr', 1e-4), betas=tuple(opt.get('betas', (0.9, 0.999))), weight_decay=opt.get('weight_decay', 0.0),
        optimizer='adamw8bit' if opt.get('type') == 'AdamW8bit' else 'adamw',       # optimizer/builder.py:11-24 registers bnb's class
        loggamma_lr_mult=mults.get('proj_out_loggamma', {}).get('lr_mult', 1.0),
        warmup_iters=lr_cfg.get('warmup_iters', 0), warmup_ratio=lr_cfg.get('warmup_ratio', 1.0),
        grad_clip=tc.get('diffusion_grad_clip', 0.0), grad_clip_begin_iter=tc.get('diffusion_grad_clip_begin_iter', 0),
        ema_gamma=ema.get('momentum_cfg', {}).get('gamma', 7.0), ema_start_iter=ema.get('start_iter', 0),
        lora_rank=den.get('lora_rank', 0) if den.get('use_lora', False) else 0,
        lora_dropout=den.get('lora_dropout', 0.0) if den.get('use_lora', False) else 0.0,
        # not in the reference's configs (it has no fp8 path): train_cfg.teacher_fp8 / student_fp8, e.g. --cfg-options train_cfg.student_fp8=True
        teacher_fp8=bool(tc.get('teacher_fp8', False)), student_fp8=bool(tc.get('student_fp8', False)))
    runner = cfg.get('runner', {})
    ck = cfg.get('checkpoint_config', {})
    run = dict(name=cfg.get('name', 'arcflow'), total_iters=cfg.get('total_iters', 10000),
               samples_per_gpu=cfg.get('data', {}).get('train_dataloader', {}).get('samples_per_gpu', 1),
               save_interval=ck.get('interval', 500), ckpt_dir=os.path.join(ck.get('out_dir', 'checkpoints/'), cfg.get('name', 'arcflow')),
               ckpt_fp16=runner.get('ckpt_fp16', False), ckpt_fp16_ema=runner.get('ckpt_fp16_ema', False),
               resume_from=cfg.get('resume_from'), load_from=cfg.get('load_from'), work_dir=cfg.get('work_dir'),
               lora_dropout=den.get('lora_dropout', 0.0), policy_kwargs=diff.get('policy_kwargs', {}),
               pretrained=den.get('pretrained'), data_train=cfg.get('data', {}).get('train', {}))
    return family, eng, dc, run
