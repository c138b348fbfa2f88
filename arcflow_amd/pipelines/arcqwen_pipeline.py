"""ArcQwenImagePipeline -- drop-in for ``lakonlab.pipelines.arcqwen_pipeline.ArcQwenImagePipeline``
(reference arcqwen_pipeline.py:65-489).  Differences from the FLUX pipeline that the reference has too:
no pooled/guidance inputs, variable text length (``prompt_embeds_mask``; only the real tokens enter the
transformer, arcqwen_pipeline.py:393 / arcqwen.py:325-330), QwenEmbedRope tables, per-channel latent
de-normalisation before the (3-D, T=1) VAE."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from ..engine import MMDiTEngine
from ..schedule import FlowMatchEulerDiscreteScheduler
from .arcflux_pipeline import _PipelineBase, load_transformer_dir


@dataclass
class QwenImagePipelineOutput:
    images: Any


class ArcQwenImagePipeline(_PipelineBase):
    _family = 'qwen'

    def __init__(self, scheduler=None, vae=None, text_encoder=None, tokenizer=None, transformer=None,
                 policy_type: str = 'ArcFlow', policy_kwargs: Optional[Dict[str, Any]] = None):
        super().__init__(scheduler, vae, text_encoder, tokenizer, transformer, policy_type, policy_kwargs)

    def _build_engine(self, num_gaussians=16, logweights_channels=4, teacher_head=False) -> MMDiTEngine:
        c = self._transformer_config
        return MMDiTEngine('qwen', c.get('num_layers', 60), 0, heads=c.get('num_attention_heads', 24),
                           head_dim=c.get('attention_head_dim', 128), in_channels=c.get('in_channels', 64),
                           joint_dim=c.get('joint_attention_dim', 3584), num_gaussians=num_gaussians,
                           logweights_channels=logweights_channels, teacher_head=teacher_head,
                           axes_dims=tuple(c.get('axes_dims_rope', (16, 56, 56))))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, torch_dtype=torch.bfloat16, **kwargs):
        root = pretrained_model_name_or_path
        if not os.path.isdir(os.path.join(root, 'transformer')):
            raise EnvironmentError(f'{root}/transformer not found: pass a local Qwen-Image snapshot directory')
        cfg, sd = load_transformer_dir(os.path.join(root, 'transformer'))
        sched_cfg = {}
        sp = os.path.join(root, 'scheduler', 'scheduler_config.json')
        if os.path.exists(sp):
            sched_cfg = json.load(open(sp))
        pipe = cls(scheduler=FlowMatchEulerDiscreteScheduler.from_config(sched_cfg))
        pipe._transformer_config, pipe._base_state_dict = cfg, sd
        if os.path.isdir(os.path.join(root, 'vae')):          # AutoencoderKLQwenImage decoder on the HIP engine
            from ..vae import AutoencoderKLQwenImageDecoder
            vcfg, vsd = load_transformer_dir(os.path.join(root, 'vae'))
            pipe.vae = AutoencoderKLQwenImageDecoder(vsd, vcfg['latents_mean'], vcfg['latents_std'],
                                                     tuple(vcfg.get('dim_mult', (1, 2, 4, 4))), vcfg.get('num_res_blocks', 2),
                                                     vcfg.get('z_dim', 16))
        if os.path.isdir(os.path.join(root, 'text_encoder')) and os.path.isdir(os.path.join(root, 'tokenizer')):
            from ..text_encoders import load_qwen25_text_encoder
            pipe.text_encoder = load_qwen25_text_encoder(os.path.join(root, 'text_encoder'))
            from transformers import AutoTokenizer
            pipe.tokenizer = AutoTokenizer.from_pretrained(os.path.join(root, 'tokenizer'))
        if 'proj_out.weight' in sd:
            pipe.transformer = pipe._build_engine(teacher_head=True)
            pipe.transformer.load_state_dict(sd)
        return pipe

    @classmethod
    def from_state_dict(cls, transformer_config, state_dict, scheduler=None, student=True, **kw):
        pipe = cls(scheduler=scheduler, **kw)
        pipe._transformer_config, pipe._base_state_dict = dict(transformer_config), state_dict
        pipe.transformer = pipe._build_engine(transformer_config.get('num_gaussians', 16),
                                              transformer_config.get('logweights_channels', 4), teacher_head=not student)
        pipe.transformer.load_state_dict(state_dict)
        return pipe

    # diffusers QwenImagePipeline prompt template (the reference inherits encode_prompt from it: arcqwen_pipeline.py:65,346)
    prompt_template_encode = ('<|im_start|>system\nDescribe the image by detailing the color, shape, size, texture, quantity, text, '
                              'spatial relationships of the objects and background:<|im_end|>\n<|im_start|>user\n{}<|im_end|>\n'
                              '<|im_start|>assistant\n')
    prompt_template_encode_start_idx = 34
    tokenizer_max_length = 1024

    def encode_prompt(self, prompt, max_sequence_length: int = 1024):
        """-> (prompt_embeds [B, T, D] zero padded, prompt_embeds_mask [B, T]): wrap in the template, run the language model,
        keep the valid tokens and drop the template's first ``prompt_template_encode_start_idx`` of them."""
        if self.text_encoder is None or self.tokenizer is None:
            raise RuntimeError('no text encoder attached: pass prompt_embeds (+ prompt_embeds_mask)')
        prompt = [prompt] if isinstance(prompt, str) else prompt
        drop = self.prompt_template_encode_start_idx
        tok = self.tokenizer([self.prompt_template_encode.format(e) for e in prompt], max_length=self.tokenizer_max_length + drop,
                             padding=True, truncation=True, return_tensors='pt')
        from ..text_encoders import Qwen25TextEncoder
        if isinstance(self.text_encoder, Qwen25TextEncoder):
            hidden = self.text_encoder(tok.input_ids, tok.attention_mask)
        else:
            hidden = self.text_encoder(input_ids=tok.input_ids.to(self.text_encoder.device), attention_mask=tok.attention_mask.to(self.text_encoder.device),
                                       output_hidden_states=True).hidden_states[-1]
        mask = tok.attention_mask.to(hidden.device).bool()
        rows = [hidden[b][mask[b]][drop:] for b in range(hidden.shape[0])]
        T = max(r.shape[0] for r in rows)
        embeds = torch.stack([torch.cat([r, r.new_zeros(T - r.shape[0], r.shape[1])]) for r in rows])
        emask = torch.stack([torch.cat([torch.ones(r.shape[0], dtype=torch.long), torch.zeros(T - r.shape[0], dtype=torch.long)]) for r in rows])
        return embeds[:, :max_sequence_length], emask[:, :max_sequence_length].to(hidden.device)

    @torch.inference_mode()
    def __call__(self, prompt: Union[str, List[str]] = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 4, total_substeps: int = 128, timestep_ratio: float = 0.5,
                 temperature: Union[float, str] = 'auto', num_images_per_prompt: int = 1,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 latents: Optional[torch.Tensor] = None, prompt_embeds: Optional[torch.Tensor] = None,
                 prompt_embeds_mask: Optional[torch.Tensor] = None, output_type: Optional[str] = 'pil',
                 return_dict: bool = True, attention_kwargs: Optional[Dict[str, Any]] = None,
                 callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ['latents'], max_sequence_length: int = 512):
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        if height % 16 or width % 16:
            raise ValueError('`height` and `width` have to be divisible by 16')
        if prompt is not None and prompt_embeds is not None:
            raise ValueError('Cannot forward both `prompt` and `prompt_embeds`.')
        if prompt_embeds is None:
            if prompt is None:
                raise ValueError('Provide either `prompt` or `prompt_embeds`.')
            prompt_embeds, prompt_embeds_mask = self.encode_prompt(prompt, max_sequence_length=max_sequence_length)
        self._apply_lora_scale(float((attention_kwargs or {}).get('scale', 1.0)))      # arcflux.py:147-154: scale_lora_layers around the forward
        if self.transformer is None or self.transformer.teacher_head:
            raise RuntimeError('load_arcflow_adapter() must be called before sampling')
        self._interrupt = False
        device = self._execution_device
        prompt_embeds = prompt_embeds.to(device, torch.bfloat16).repeat_interleave(num_images_per_prompt, dim=0)
        if prompt_embeds_mask is not None:
            lens = prompt_embeds_mask.sum(dim=1).tolist()          # the reference's host sync (arcqwen_pipeline.py:393)
            prompt_embeds = prompt_embeds[:, :int(max(lens))]
        B = prompt_embeds.shape[0]
        latents, hp, wp = self._prepare_latents(B, height, width, generator, latents)

        def fwd(x, t, pe, prepared_step=None):
            return self.transformer(x, t, pe.to(device, torch.bfloat16), None, None, hp, wp, prepared_step=prepared_step)

        def prepare(sigmas):
            return self.transformer.prepare_steps(sigmas, None, None, B, hp * wp, prompt_embeds.shape[1])
        latents = self._denoise(latents, hp, wp, num_inference_steps, total_substeps, timestep_ratio, fwd,
                                callback_on_step_end, callback_on_step_end_tensor_inputs, prompt_embeds, prepare)
        if output_type == 'latent':
            image = latents
        else:
            if self.vae is None:
                raise RuntimeError("no VAE decoder attached: use output_type='latent' or set pipe.vae")
            from ..vae import AutoencoderKLQwenImageDecoder
            if isinstance(self.vae, AutoencoderKLQwenImageDecoder):   # HIP decoder: un-normalisation + unpack fused into its first kernel
                image = self._postprocess(self.vae.decode_packed(latents, hp, wp), output_type)
                return (image,) if not return_dict else QwenImagePipelineOutput(images=image)
            lat = self._unpack(latents, hp, wp)[:, :, None]
            mean = torch.tensor(self.vae.config.latents_mean, device=device).view(1, -1, 1, 1, 1)
            std = torch.tensor(self.vae.config.latents_std, device=device).view(1, -1, 1, 1, 1)
            image = self.vae.decode((lat * std + mean).to(next(self.vae.parameters()).dtype), return_dict=False)[0][:, :, 0]
            image = self._postprocess(image, output_type)
        if not return_dict:
            return (image,)
        return QwenImagePipelineOutput(images=image)
