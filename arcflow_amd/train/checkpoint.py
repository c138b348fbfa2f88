"""Checkpoint / adapter interchange with the reference's on-disk formats.

Two layouts are read and written:

* ``iter_N.pth`` -- the runner checkpoint (lakonlab/runner/dynamic_iter_based_runner.py:106-158,
  lakonlab/runner/checkpoint.py:491-534): ``{'meta': {'iter', 'epoch', ...}, 'state_dict': {...}, 'optimizer': {...}}``
  where the state dict holds the TRAINABLE tensors only, live copy under ``diffusion.denoising.<name>`` and the
  EMA copy under ``diffusion_ema.denoising.<name>``, LoRA matrices with their peft names
  (``<module>.lora_A.default.weight``), non-EMA tensors stored fp16 when ``ckpt_fp16`` is set.
* the adapter directory (export_arcflow_to_diffusers.py:100-127): ``config.json`` with ``_class_name`` and the
  constructor arguments, ``diffusion_pytorch_model.safetensors`` with the prefix stripped and
  ``lora_A.default.weight -> lora_A.weight``, metadata ``policy_config`` (JSON, ``type='ArcFlow'``).
  ``ArcFlowLoaderMixin.load_arcflow_adapter`` (pipelines/arcflow_loader.py) ingests exactly this.

The optimizer entry is this engine's flat fp32 AdamW state; the reference stores a bitsandbytes 8-bit state there,
which is not interchangeable -- loading a reference checkpoint restores weights + EMA and restarts the moments.
"""
from __future__ import annotations

import json
import os
import threading
import time
import warnings
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

LIVE_PREFIX = 'diffusion.denoising.'
EMA_PREFIX = 'diffusion_ema.denoising.'
CLASS_NAMES = {'flux': 'ArcFluxTransformer2DModel', 'qwen': 'ArcQwenImageTransformer2DModel'}
SAFETENSORS_WEIGHTS_NAME = 'diffusion_pytorch_model.safetensors'


@dataclass
class TrainableLayout:
    """Where every trainable tensor lives inside the distiller's flat fp32 buffer."""
    K: int
    C: int
    L: int
    D: int
    head_n: int                    # stacked + padded head rows (means | logweights | loggamma | pad)
    offsets: List[int]             # [head.weight, head.bias, norm_out.weight, norm_out.bias, lora...] starts, then the end
    rank: int
    lora: List[Tuple[str, int, int, int, int]]     # (peft module name, out_f, in_f, off_a, off_b)

    @property
    def numel(self) -> int:
        return self.lora[-1][4] + self.lora[-1][1] * self.rank if self.lora else self.offsets[4]


def flat_to_state(flat: torch.Tensor, lay: TrainableLayout, peft_names: bool = False) -> Dict[str, torch.Tensor]:
    """Flat fp32 buffer -> tensors under the reference's module names (views; clone before mutating)."""
    K, C, L, D = lay.K, lay.C, lay.L, lay.D
    o = lay.offsets
    hw = flat[o[0]:o[1]].view(lay.head_n, D)
    hb = flat[o[1]:o[2]]
    n1, n2 = K * C, K * C + K * L
    n3 = n2 + (K - 1) * L
    sd = {
        'proj_out_means.weight': hw[:n1], 'proj_out_means.bias': hb[:n1],
        'proj_out_logweights.weight': hw[n1:n2], 'proj_out_logweights.bias': hb[n1:n2],
        'proj_out_loggamma.weight': hw[n2:n3], 'proj_out_loggamma.bias': hb[n2:n3],
        'norm_out.linear.weight': flat[o[2]:o[3]].view(2 * D, D), 'norm_out.linear.bias': flat[o[3]:o[4]],
    }
    mid = '.default' if peft_names else ''
    for name, out_f, in_f, off_a, off_b in lay.lora:
        sd[f'{name}.lora_A{mid}.weight'] = flat[off_a:off_a + lay.rank * in_f].view(lay.rank, in_f)
        sd[f'{name}.lora_B{mid}.weight'] = flat[off_b:off_b + out_f * lay.rank].view(out_f, lay.rank)
    return sd


def state_to_flat(sd: Dict[str, torch.Tensor], flat: torch.Tensor, lay: TrainableLayout, strict: bool = True) -> List[str]:
    """Copy a (reference- or export-named) trainable state dict into the flat buffer.  Returns the keys of ``sd`` that
    this engine does not train (adapters on other modules); a missing trainable raises when ``strict``."""
    norm = {k.replace('.lora_A.default.', '.lora_A.').replace('.lora_B.default.', '.lora_B.'): v for k, v in sd.items()}
    dst = flat_to_state(flat, lay, peft_names=False)
    missing = [k for k in dst if k not in norm]
    if missing and strict:
        raise KeyError(f'checkpoint lacks trainable tensors: {missing[:4]}{" ..." if len(missing) > 4 else ""}')
    for k, d in dst.items():
        if k in norm:
            s = norm[k]
            if tuple(s.shape) != tuple(d.shape):
                raise ValueError(f'{k}: checkpoint shape {tuple(s.shape)} != expected {tuple(d.shape)}')
            d.copy_(s.to(device=d.device, dtype=d.dtype))
    return sorted(k for k in norm if k not in dst)


def layout_of(distiller) -> TrainableLayout:
    lora = []
    rank = 0
    if distiller.trunk is not None:
        rank = distiller.trunk.r
        lora = [(sp.name, sp.out_f, sp.in_f, sp.off_a, sp.off_b) for sp in distiller.trunk.specs]
    return TrainableLayout(distiller.K, distiller.C, distiller.L, distiller.D, distiller.head_n, list(distiller._off), rank, lora)


# ---------------------------------------------------------------------------------------------- iter_N.pth
def build_checkpoint(distiller, fp16: bool = True, fp16_ema: bool = False, save_optimizer: bool = True, meta: Optional[dict] = None) -> dict:
    lay = layout_of(distiller)
    state = {}
    for prefix, flat, half in ((LIVE_PREFIX, distiller.params, fp16), (EMA_PREFIX, distiller.ema, fp16_ema)):
        for k, v in flat_to_state(flat, lay, peft_names=True).items():
            v = v.detach()
            state[prefix + k] = (v.half() if half else v).to('cpu', copy=True)      # (rounded on the device: half the transfer, no host-side conversion loop)
    m = dict(meta or {})
    m.update(iter=distiller.iteration, epoch=1, time=time.asctime(), writer='arcflow_amd')
    ckpt = {'meta': m, 'state_dict': state}
    if save_optimizer and getattr(distiller, 'exp_avg', None) is None:       # adamw8bit: per learning-rate group, codes + block absmax
        ckpt['optimizer'] = {'diffusion': {'format': 'arcflow_amd.flat_adamw8bit', 'step': distiller.opt_steps,
                                           'groups8': {k: st.state_dict() for k, st in distiller.opt8.items()},
                                           'groups32': {k: (m.detach().cpu(), v.detach().cpu()) for k, (m, v) in distiller._small.items()}}}
    elif save_optimizer:
        ckpt['optimizer'] = {'diffusion': {'format': 'arcflow_amd.flat_adamw', 'step': distiller.opt_steps,
                                           'exp_avg': distiller.exp_avg.detach().cpu(), 'exp_avg_sq': distiller.exp_avg_sq.detach().cpu()}}
    return ckpt


def save_checkpoint(distiller, out_dir: str, filename_tmpl: str = 'iter_{}.pth', create_symlink: bool = True, **kw) -> str:
    """Rank-0 style save (the caller decides which rank writes)."""
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, filename_tmpl.format(distiller.iteration))
    with open(path, 'wb') as f:
        torch.save(build_checkpoint(distiller, **kw), f)
        f.flush()
    if create_symlink:
        link = os.path.join(out_dir, 'latest.pth')
        if os.path.lexists(link):
            os.remove(link)
        os.symlink(os.path.basename(path), link)
    return path


_pending_save: Optional[threading.Thread] = None
_pending_error: List[BaseException] = []


def wait_pending_save() -> None:
    """Block until a background save started by ``save_checkpoint_async`` has reached the disk (re-raises its error)."""
    global _pending_save
    if _pending_save is not None:
        _pending_save.join()
        _pending_save = None
    if _pending_error:
        raise _pending_error.pop()


def check_pending_save() -> None:
    """Cheap, non-blocking: re-raise the error of a background save that has already failed (tools/train.py calls it every iteration, so a full
    disk stops the run at once instead of at the next save_interval)."""
    if _pending_error:
        raise _pending_error.pop()


def pending_save_failed() -> bool:
    """True when a background save has failed and its error has not been raised yet (tools/train.py shares this flag across ranks before raising)."""
    return bool(_pending_error)


def save_checkpoint_async(distiller, out_dir: str, filename_tmpl: str = 'iter_{}.pth', create_symlink: bool = True, **kw) -> str:
    """``save_checkpoint`` with the file write off the training loop: the state is copied to host memory here (that part synchronises the device:
    ~1 s for the Qwen-Image adapter set), pickling and the ~6 GB write run in a thread.  The file appears under its final name only when complete
    (written as ``*.tmp``, then renamed), ``latest.pth`` moves after that; a second save waits for the first; call ``wait_pending_save()`` before exit."""
    global _pending_save
    wait_pending_save()
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, filename_tmpl.format(distiller.iteration))
    ckpt = build_checkpoint(distiller, **kw)

    def work():
        tmp = path + '.tmp'
        try:
            with open(tmp, 'wb') as f:
                torch.save(ckpt, f)
                f.flush()
                os.fsync(f.fileno())            # the data is on the disk before the final name exists
            os.replace(tmp, path)
            if create_symlink:                  # latest.pth moves atomically too: a kill between "remove" and "symlink" would leave a run without it
                link, tmp_link = os.path.join(out_dir, 'latest.pth'), os.path.join(out_dir, '.latest.pth.tmp')
                if os.path.lexists(tmp_link):
                    os.remove(tmp_link)
                os.symlink(os.path.basename(path), tmp_link)
                os.replace(tmp_link, link)
        except BaseException as e:      # noqa: BLE001  (handed to the training thread by wait_pending_save / check_pending_save)
            try:
                if os.path.exists(tmp):
                    os.remove(tmp)              # no multi-GB *.tmp left behind by a failed write
            except OSError:
                pass
            _pending_error.append(e)

    _pending_save = threading.Thread(target=work, name='arcflow-checkpoint-writer')
    _pending_save.start()
    return path


def load_checkpoint(distiller, path: str, strict: bool = True) -> dict:
    """Restore weights, EMA, moments and the iteration counter (runner.resume semantics).  Returns the meta dict."""
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    sd = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    lay = layout_of(distiller)
    live = {k[len(LIVE_PREFIX):]: v for k, v in sd.items() if k.startswith(LIVE_PREFIX)}
    ema = {k[len(EMA_PREFIX):]: v for k, v in sd.items() if k.startswith(EMA_PREFIX)}
    if not live and not ema:
        raise KeyError(f'{path}: no {LIVE_PREFIX}* / {EMA_PREFIX}* tensors')
    if not live:
        live = ema
    extra = state_to_flat(live, distiller.params, lay, strict)
    if ema:
        state_to_flat(ema, distiller.ema, lay, strict)
    else:
        distiller.ema.copy_(distiller.params)
    if extra:
        warnings.warn(f'{len(extra)} checkpoint tensors are not trained by this engine and were skipped: {extra[:4]}')
    opt = (ckpt.get('optimizer') or {}).get('diffusion')
    if getattr(distiller, 'exp_avg', None) is None:                          # adamw8bit distiller
        from .. import ops
        distiller.opt8.clear()
        distiller._small.clear()
        distiller.opt_steps = 0
        want = {(a, b) for a, b, _ in distiller.optimizer_groups()}
        have = set(opt.get('groups8', {})) | set(opt.get('groups32', {})) if isinstance(opt, dict) else set()
        if isinstance(opt, dict) and opt.get('format') == 'arcflow_amd.flat_adamw8bit' and not have <= want:
            # written under another layout (lora_rank, block count, pre-round-3 per-learning-rate grouping): the (a, b) ranges do not
            # describe this distiller's tensors -- restoring them would pair moments with the wrong parameters
            warnings.warn(f'{path}: 8-bit optimizer state was written for a different parameter layout '
                          f'({len(have - want)} of {len(have)} ranges unknown here): moments restart at zero')
        elif isinstance(opt, dict) and opt.get('format') == 'arcflow_amd.flat_adamw8bit':
            for (a, b), sd in opt['groups8'].items():
                st = ops.AdamW8bitState(b - a, distiller.device)
                st.load_state_dict(sd)
                distiller.opt8[(a, b)] = st
            for k, (m, v) in opt['groups32'].items():
                distiller._small[k] = (m.to(distiller.device), v.to(distiller.device))
            distiller.opt_steps = int(opt['step'])
        elif 'optimizer' in ckpt:
            warnings.warn('optimizer state is not this engine\'s 8-bit format: moments restart at zero')
    elif isinstance(opt, dict) and opt.get('format') == 'arcflow_amd.flat_adamw' and opt['exp_avg'].numel() == distiller.params.numel():
        distiller.exp_avg.copy_(opt['exp_avg'])
        distiller.exp_avg_sq.copy_(opt['exp_avg_sq'])
        distiller.opt_steps = int(opt['step'])
    else:
        if 'optimizer' in ckpt:
            warnings.warn('optimizer state is not in this engine\'s format (bitsandbytes 8-bit state of the reference?): moments restart at zero')
        distiller.exp_avg.zero_()
        distiller.exp_avg_sq.zero_()
        distiller.opt_steps = 0
    meta = ckpt.get('meta', {})
    distiller.iteration = int(meta.get('iter', 0))
    distiller._sync_working_copies()
    if distiller.trunk is not None:
        distiller.trunk.refresh()
    return meta


# ---------------------------------------------------------------------------------------------- adapter directory
def adapter_config(family: str, engine, lora_rank: int) -> dict:
    """The constructor arguments the reference's ``save_config`` would dump (export_arcflow_to_diffusers.py:44-58)."""
    cfg = {'_class_name': CLASS_NAMES[family], '_diffusers_version': '0.35.1', 'patch_size': 2,
           'num_gaussians': engine.num_gaussians, 'logweights_channels': engine.logweights_channels,
           'in_channels': engine.in_channels, 'num_layers': engine.num_double,
           'attention_head_dim': engine.dim // engine.heads, 'num_attention_heads': engine.heads,
           'joint_attention_dim': engine.joint_dim}
    if family == 'flux':
        cfg.update(num_single_layers=engine.num_single, pooled_projection_dim=engine.pooled_dim,
                   guidance_embeds=bool(engine.guidance_embeds), axes_dims_rope=[16, 56, 56])
    else:
        cfg.update(out_channels=16, axes_dims_rope=[16, 56, 56])
    return cfg


def export_adapter_state(state: Dict[str, torch.Tensor], ema: bool = True) -> Dict[str, torch.Tensor]:
    """Runner state dict -> adapter tensors: pick the live or EMA copy, strip the prefix, drop peft's adapter name."""
    prefix = EMA_PREFIX if ema else LIVE_PREFIX
    out = {}
    for k, v in state.items():
        if k.startswith(prefix):
            nk = k[len(prefix):].replace('lora_A.default.weight', 'lora_A.weight').replace('lora_B.default.weight', 'lora_B.weight')
            out[nk] = v.contiguous()
    return out


def export_adapter(distiller, out_dir: str, ema: bool = True, policy_kwargs: Optional[dict] = None, dtype: Optional[torch.dtype] = None) -> str:
    """Write the diffusers-style adapter directory that ``load_arcflow_adapter`` (here and in the reference) reads."""
    from safetensors.torch import save_file
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, 'config.json'), 'w', encoding='utf-8') as f:
        f.write(json.dumps(adapter_config(distiller.family, distiller.student, distiller.cfg.lora_rank), indent=2, sort_keys=True) + '\n')
    ck = build_checkpoint(distiller, fp16=False, fp16_ema=False, save_optimizer=False)
    tensors = export_adapter_state(ck['state_dict'], ema=ema)
    if dtype is not None:
        tensors = {k: v.to(dtype) for k, v in tensors.items()}
    policy = dict(policy_kwargs or {})
    policy.update(type='ArcFlow')
    save_file(tensors, os.path.join(out_dir, SAFETENSORS_WEIGHTS_NAME), metadata={'policy_config': json.dumps(policy)})
    return out_dir
