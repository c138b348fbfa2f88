"""``load_arcflow_adapter()`` with the reference's contract
(lakonlab/pipelines/arcflow_loader.py:45-275): read ``config.json`` (``_class_name`` must be an ArcFlow
transformer), read ``diffusion_pytorch_model.safetensors`` (keys written by
export_arcflow_to_diffusers.py:100-127), overlay every non-LoRA key (three heads, ``norm_out``) on the
base transformer's weights, fold the LoRA pairs in, swap the pipeline's ``transformer`` for the ArcFlow
student and return ``f"{target_module_name}_arcflow"`` (or ``None`` + a warning without LoRA keys).

MI355X specifics: the student is a new ``MMDiTEngine`` with the 3-head velocity output; LoRA is merged
into the bf16 base weights at load (fp32 merge, one rounding) instead of running side GEMMs per call.
"""
from __future__ import annotations

import json
import os
import warnings
from typing import Dict, Optional

import torch

LOCAL_CLASS_MAPPING = {
    'ArcFluxTransformer2DModel': 'flux',
    'ArcQwenImageTransformer2DModel': 'qwen',
}
SAFETENSORS_WEIGHTS_NAME = 'diffusion_pytorch_model.safetensors'
_HF_KWARGS = ('cache_dir', 'force_download', 'proxies', 'token', 'local_files_only', 'revision', 'subfolder',
              'low_cpu_mem_usage', 'variant', 'use_safetensors', 'disable_mmap')


def _resolve_dir(path: str, subfolder: Optional[str]) -> str:
    d = os.path.join(path, subfolder) if subfolder else path
    if not os.path.isdir(d):
        raise EnvironmentError(
            f'{d} is not a local directory. This build has no network access: pass a local snapshot of the '
            f'adapter repository (the layout export_arcflow_to_diffusers.py writes).')
    return d


def read_adapter(path: str, subfolder: Optional[str] = None, variant: Optional[str] = None):
    """-> (config dict, state dict, safetensors metadata)."""
    from safetensors import safe_open
    d = _resolve_dir(path, subfolder)
    with open(os.path.join(d, 'config.json')) as f:
        config = json.load(f)
    name = SAFETENSORS_WEIGHTS_NAME if not variant else SAFETENSORS_WEIGHTS_NAME.replace('.safetensors', f'.{variant}.safetensors')
    sd: Dict[str, torch.Tensor] = {}
    with safe_open(os.path.join(d, name), framework='pt', device='cpu') as f:
        meta = f.metadata() or {}
        for k in f.keys():
            sd[k] = f.get_tensor(k)
    return config, sd, meta


class ArcFlowLoaderMixin:
    """Adds ``load_arcflow_adapter`` to a pipeline that keeps ``self._base_state_dict`` (diffusers keys)
    and ``self._transformer_config``."""

    def load_arcflow_adapter(self, pretrained_model_name_or_path: str, target_module_name: str = 'transformer',
                             adapter_name: Optional[str] = None, **kwargs) -> Optional[str]:
        unknown = set(kwargs) - set(_HF_KWARGS)
        if unknown:
            raise TypeError(f'load_arcflow_adapter() got unexpected keyword arguments {sorted(unknown)}')
        subfolder = kwargs.get('subfolder')
        config, adapter_sd, meta = read_adapter(pretrained_model_name_or_path, subfolder, kwargs.get('variant'))
        cls_name = config.get('_class_name')
        if cls_name not in LOCAL_CLASS_MAPPING:
            raise ValueError(f"Can't find a model linked to {cls_name}.")
        family = LOCAL_CLASS_MAPPING[cls_name]
        if family != self._family:
            raise ValueError(f'{cls_name} adapter cannot be loaded into a {self._family} pipeline')
        base = dict(self._base_state_dict)
        lora: Dict[str, torch.Tensor] = {}
        prefix = target_module_name + '.'
        for k, v in adapter_sd.items():
            k2 = k[len(prefix):] if k.startswith(prefix) else k
            (lora if 'lora' in k2 else base)[k2] = v
        if len(lora) == 0:
            warnings.warn(f'No LoRA weights were found in {pretrained_model_name_or_path}.')
            return None
        if adapter_name is None:
            adapter_name = f'{target_module_name}_arcflow'
        from ..weights import merge_lora
        merged = merge_lora(base, lora, scale=1.0)
        student = self._build_engine(num_gaussians=config.get('num_gaussians', 16),
                                     logweights_channels=config.get('logweights_channels', 4), teacher_head=False)
        student.load_state_dict(merged)
        setattr(self, target_module_name, student)
        self.policy_config = json.loads(meta['policy_config']) if 'policy_config' in meta else {'type': 'ArcFlow'}
        self._adapters = getattr(self, '_adapters', []) + [adapter_name]
        # kept for a runtime LoRA scale (`joint_attention_kwargs={'scale': s}` / `set_adapters(..., adapter_weights=s)`): see _apply_lora_scale
        self._adapter_state = dict(target=target_module_name, base=base, lora=lora, merged_scale=1.0, weight=1.0)
        return adapter_name

    # ------------------------------------------------------------------ runtime LoRA scale
    def set_adapters(self, adapter_names, adapter_weights=None) -> None:
        """diffusers' `pipe.set_adapters(names, adapter_weights=w)` for the ArcFlow adapter (inference_flux.py:9 mentions it): the adapter's
        LoRA branch is weighted by w from now on.  Only the loaded ArcFlow adapter is known here (style LoRAs would be further
        `lora` dicts folded the same way)."""
        names = [adapter_names] if isinstance(adapter_names, str) else list(adapter_names)
        known = getattr(self, '_adapters', [])
        for n in names:
            if n not in known:
                raise ValueError(f'adapter {n!r} is not loaded (loaded: {known})')
        if adapter_weights is None:
            w = 1.0
        elif isinstance(adapter_weights, (int, float)):
            w = float(adapter_weights)
        else:
            w = float(list(adapter_weights)[0])
        self._adapter_state['weight'] = w
        self._apply_lora_scale(1.0)

    def _apply_lora_scale(self, call_scale: float) -> None:
        """The reference scales every LoRA layer by `joint_attention_kwargs['scale']` around the forward (`scale_lora_layers` /
        `unscale_lora_layers`, lakonlab/models/architecture/arcflow/arcflux.py:147-154, :251-252): y = W x + s (alpha / r) B A x.  Here the
        adapter is folded into the weights, so a different s means folding again: W + s B A from the kept base and LoRA tensors (one
        fp32-accumulating GEMM per adapted linear on the device + a re-pack, about a second for FLUX-12B), cached until s changes --
        a call with the same scale as the previous one costs nothing.  Deviation of the folded forward from the un-folded one:
        tests/test_distill.py::test_unmerged_trunk_forward_matches_merged_engine."""
        st = getattr(self, '_adapter_state', None)
        if st is None:
            if call_scale != 1.0:
                raise RuntimeError('a LoRA scale was passed but no ArcFlow adapter is loaded')
            return
        s = float(call_scale) * st['weight']
        if s == st['merged_scale']:
            return
        from ..weights import merge_lora
        getattr(self, st['target']).load_state_dict(merge_lora(st['base'], st['lora'], scale=s))
        st['merged_scale'] = s
