"""ArcFluxPipeline -- drop-in for ``lakonlab.pipelines.arcflux_pipeline.ArcFluxPipeline``
(reference arcflux_pipeline.py:73-542): same constructor extras (``policy_type``, ``policy_kwargs``),
same ``__call__`` keywords and defaults, ``.images`` result, ``load_arcflow_adapter``.

What runs where: the denoising loop (arcflux_pipeline.py:457-510) is two C-ABI calls per step --
``afx_mmdit_forward`` and ``afx_arcflow_step`` -- on latents that never leave the packed token layout.
Prompt encoding and VAE decode are the rows SURVEY 8f marks "next": pass ``prompt_embeds`` /
``pooled_prompt_embeds`` (or attach HF text encoders) and use ``output_type='latent'`` unless a decoder
module is attached as ``pipe.vae``.
"""
from __future__ import annotations

import glob
import json
import os
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from .. import ops
from ..engine import MMDiTEngine
from ..schedule import FlowMatchEulerDiscreteScheduler, calculate_shift, retrieve_raw_timesteps
from .arcflow_loader import ArcFlowLoaderMixin

POLICY_CLASSES = ('ArcFlow',)


@dataclass
class FluxPipelineOutput:
    images: Any


def load_transformer_dir(path: str):
    """Read ``<path>/config.json`` + every ``*.safetensors`` shard of a diffusers transformer folder."""
    from safetensors import safe_open
    with open(os.path.join(path, 'config.json')) as f:
        cfg = json.load(f)
    sd: Dict[str, torch.Tensor] = {}
    files = sorted(glob.glob(os.path.join(path, '*.safetensors')))
    if not files:
        raise EnvironmentError(f'no safetensors weights under {path}')
    for fn in files:
        with safe_open(fn, framework='pt', device='cpu') as f:
            for k in f.keys():
                sd[k] = f.get_tensor(k)
    return cfg, sd


class _PipelineBase(ArcFlowLoaderMixin):
    _family = 'flux'
    vae_scale_factor = 8
    default_sample_size = 128

    def __init__(self, scheduler=None, vae=None, text_encoder=None, tokenizer=None, transformer=None,
                 policy_type: str = 'ArcFlow', policy_kwargs: Optional[Dict[str, Any]] = None):
        assert policy_type in POLICY_CLASSES, \
            f'Invalid policy: {policy_type}. Supported policies are {list(POLICY_CLASSES)}.'
        self.policy_type = policy_type
        self.policy_kwargs = policy_kwargs or {}
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler(shift=3.0, use_dynamic_shifting=True)
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.transformer = transformer
        self._base_state_dict: Dict[str, torch.Tensor] = {}
        self._transformer_config: Dict[str, Any] = {}
        self._device = torch.device('cuda')
        self._interrupt = False
        self._num_timesteps = 0
        self._current_timestep = None

    # diffusers-style plumbing -------------------------------------------------------------------
    def to(self, device=None, *a, **k):
        if device is not None and torch.device(device).type != 'cuda':
            raise RuntimeError('arcflow_amd pipelines run on the GPU only (no CPU fallback)')
        return self

    def enable_model_cpu_offload(self, *a, **k):      # 288 GB HBM: nothing to offload
        return self

    def maybe_free_model_hooks(self):
        pass

    @property
    def interrupt(self):
        return self._interrupt

    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def _execution_device(self):
        return self._device

    def _retrieve_timesteps(self, raw, device, image_seq_len):
        mu = calculate_shift(image_seq_len, self.scheduler.config.get('base_image_seq_len', 256),
                             self.scheduler.config.get('max_image_seq_len', 4096),
                             self.scheduler.config.get('base_shift', 0.5), self.scheduler.config.get('max_shift', 1.15))
        self.scheduler.set_timesteps(sigmas=raw, device=device, mu=mu)
        return self.scheduler.timesteps

    def _prepare_latents(self, batch, height, width, generator, latents):
        hp, wp = int(height) // (self.vae_scale_factor * 2), int(width) // (self.vae_scale_factor * 2)
        if latents is not None:
            return latents.to(self._device, torch.float32), hp, wp
        shape = (batch, 16, 2 * hp, 2 * wp)
        if isinstance(generator, list):
            noise = torch.cat([torch.randn((1,) + shape[1:], generator=g, device=g.device if hasattr(g, 'device') else 'cpu',
                                           dtype=torch.float32).to(self._device) for g in generator])
        else:
            gdev = generator.device if generator is not None else self._device
            noise = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32).to(self._device)
        # FLUX packing: [B,16,2hp,2wp] -> [B, hp*wp, 64], channel = c*4 + ph*2 + pw (arcflux_pipeline.py:162-175)
        packed = noise.view(batch, 16, hp, 2, wp, 2).permute(0, 2, 4, 1, 3, 5).reshape(batch, hp * wp, 64)
        return packed.contiguous(), hp, wp

    def _denoise(self, latents, hp, wp, num_inference_steps, total_substeps, timestep_ratio, fwd,
                 callback_on_step_end, callback_on_step_end_tensor_inputs, prompt_embeds, prepare=None):
        device = self._device
        raw, per_step, total = retrieve_raw_timesteps(num_inference_steps, total_substeps, timestep_ratio)
        timesteps = self._retrieve_timesteps(raw, device, latents.shape[1])
        assert len(timesteps) == total
        self._num_timesteps = total
        self.scheduler.set_begin_index(0)
        ntt = self.scheduler.config.num_train_timesteps
        ts_host = timesteps.float().cpu().tolist()          # 128 floats, once per call
        # every step's source timestep is known here: the AdaLN modulation vectors of ALL steps (functions of t, guidance and the
        # pooled text only) come out of one pass over the stacked modulation matrix instead of one pass per transformer call
        starts = [sum(per_step[:i]) for i in range(num_inference_steps)]
        prepared = bool(prepare([ts_host[j] / 1000.0 for j in starts])) if prepare is not None and not self.interrupt else False
        tid = 0
        for i in range(num_inference_steps):
            if self.interrupt:
                continue
            t_src = ts_host[tid]
            sigma_src = t_src / ntt
            self._current_timestep = t_src
            out = fwd(latents.to(torch.bfloat16), torch.full((latents.shape[0],), t_src / 1000.0, device=device), prompt_embeds,
                      i if prepared else None)
            tid += per_step[i]
            sigma_end = (ts_host[tid] / ntt) if tid < len(ts_host) else 0.0
            latents = ops.arcflow_step(latents, out.means, out.logweights, out.loggammas,
                                       sigma_src, sigma_src, sigma_end, eps=1e-4)
            if callback_on_step_end is not None:
                local = dict(latents=latents, prompt_embeds=prompt_embeds)
                cb = callback_on_step_end(self, i, torch.tensor(t_src, device=device),
                                          {k: local[k] for k in callback_on_step_end_tensor_inputs})
                latents = cb.pop('latents', latents)
                # the reference also takes back a modified conditioning (arcflux_pipeline.py:519 pops "prompt_embeds")
                prompt_embeds = cb.pop('prompt_embeds', prompt_embeds)
        self._current_timestep = None
        return latents

    def _unpack(self, latents, hp, wp):
        b = latents.shape[0]
        return latents.view(b, hp, wp, 16, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(b, 16, 2 * hp, 2 * wp)

    def _postprocess(self, image, output_type):
        if output_type == 'pt':
            return image
        img = (image.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()
        if output_type == 'np':
            return img
        from PIL import Image
        return [Image.fromarray((x * 255).round().astype('uint8')) for x in img]


class ArcFluxPipeline(_PipelineBase):
    r"""Policy-based FLUX pipeline, 2-NFE capable (reference arcflux_pipeline.py:73)."""
    _family = 'flux'

    def __init__(self, scheduler=None, vae=None, text_encoder=None, tokenizer=None, text_encoder_2=None,
                 tokenizer_2=None, transformer=None, image_encoder=None, feature_extractor=None,
                 policy_type: str = 'ArcFlow', policy_kwargs: Optional[Dict[str, Any]] = None):
        super().__init__(scheduler, vae, text_encoder, tokenizer, transformer, policy_type, policy_kwargs)
        self.text_encoder_2, self.tokenizer_2 = text_encoder_2, tokenizer_2

    # construction -----------------------------------------------------------------------------------
    def _build_engine(self, num_gaussians=16, logweights_channels=4, teacher_head=False) -> MMDiTEngine:
        c = self._transformer_config
        return MMDiTEngine('flux', c.get('num_layers', 19), c.get('num_single_layers', 38),
                           heads=c.get('num_attention_heads', 24), head_dim=c.get('attention_head_dim', 128),
                           in_channels=c.get('in_channels', 64), joint_dim=c.get('joint_attention_dim', 4096),
                           pooled_dim=c.get('pooled_projection_dim', 768), guidance_embeds=c.get('guidance_embeds', True),
                           num_gaussians=num_gaussians, logweights_channels=logweights_channels,
                           teacher_head=teacher_head, axes_dims=tuple(c.get('axes_dims_rope', (16, 56, 56))))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, torch_dtype=torch.bfloat16, **kwargs):
        """Local diffusers snapshot (``transformer/``, ``scheduler/`` ...).  No network in this build."""
        if torch_dtype not in (torch.bfloat16, None):
            raise ValueError('the MI355X engine computes in bf16')
        root = pretrained_model_name_or_path
        if not os.path.isdir(os.path.join(root, 'transformer')):
            raise EnvironmentError(f'{root}/transformer not found: pass a local FLUX.1-dev snapshot directory')
        cfg, sd = load_transformer_dir(os.path.join(root, 'transformer'))
        sched_cfg = {}
        sp = os.path.join(root, 'scheduler', 'scheduler_config.json')
        if os.path.exists(sp):
            sched_cfg = json.load(open(sp))
        pipe = cls(scheduler=FlowMatchEulerDiscreteScheduler.from_config(sched_cfg))
        pipe._transformer_config, pipe._base_state_dict = cfg, sd
        if os.path.isdir(os.path.join(root, 'vae')):          # AutoencoderKL decoder on the HIP engine
            from ..vae import AutoencoderKLDecoder
            vcfg, vsd = load_transformer_dir(os.path.join(root, 'vae'))
            pipe.vae = AutoencoderKLDecoder(vsd, tuple(vcfg.get('block_out_channels', (128, 256, 512, 512))),
                                            vcfg.get('norm_num_groups', 32), vcfg.get('layers_per_block', 2),
                                            vcfg.get('scaling_factor', 0.3611), vcfg.get('shift_factor', 0.1159))
        if os.path.isdir(os.path.join(root, 'text_encoder')) and os.path.isdir(os.path.join(root, 'text_encoder_2')):
            from ..text_encoders import load_clip_text_encoder, load_t5_encoder       # prompt encoders on the HIP engine
            pipe.text_encoder = load_clip_text_encoder(os.path.join(root, 'text_encoder'))
            pipe.text_encoder_2 = load_t5_encoder(os.path.join(root, 'text_encoder_2'))
            from transformers import AutoTokenizer                                      # tokenisation is host-side plumbing
            pipe.tokenizer = AutoTokenizer.from_pretrained(os.path.join(root, 'tokenizer'))
            pipe.tokenizer_2 = AutoTokenizer.from_pretrained(os.path.join(root, 'tokenizer_2'))
        if 'proj_out.weight' in sd:          # plain FLUX: usable as the teacher until an adapter is loaded
            pipe.transformer = pipe._build_engine(teacher_head=True)
            pipe.transformer.load_state_dict(sd)
        return pipe

    @classmethod
    def from_state_dict(cls, transformer_config: Dict[str, Any], state_dict: Dict[str, torch.Tensor],
                        scheduler=None, student: bool = True, **kw):
        """Build from an in-memory diffusers-keyed state dict (tests, synthetic weights)."""
        pipe = cls(scheduler=scheduler, **kw)
        pipe._transformer_config, pipe._base_state_dict = dict(transformer_config), state_dict
        pipe.transformer = pipe._build_engine(transformer_config.get('num_gaussians', 16),
                                              transformer_config.get('logweights_channels', 4), teacher_head=not student)
        pipe.transformer.load_state_dict(state_dict)
        return pipe

    # prompt encoding (text encoders are a "next" row) -------------------------------------
    def encode_prompt(self, prompt, prompt_2, prompt_embeds, pooled_prompt_embeds, device, num_images_per_prompt,
                      max_sequence_length):
        if prompt_embeds is None:
            if self.text_encoder is None or self.text_encoder_2 is None:
                raise RuntimeError('no text encoders attached: pass prompt_embeds and pooled_prompt_embeds')
            prompt = [prompt] if isinstance(prompt, str) else prompt
            prompt_2 = prompt if prompt_2 is None else ([prompt_2] if isinstance(prompt_2, str) else prompt_2)
            from ..text_encoders import CLIPTextEncoder, T5Encoder
            with torch.no_grad():      # diffusers FluxPipeline._get_clip_prompt_embeds / _get_t5_prompt_embeds: no attention masks
                ids = self.tokenizer(prompt, padding='max_length', max_length=77, truncation=True, return_tensors='pt').input_ids
                if isinstance(self.text_encoder, CLIPTextEncoder):
                    pooled_prompt_embeds = self.text_encoder(ids)[1]
                else:
                    pooled_prompt_embeds = self.text_encoder(ids.to(self.text_encoder.device)).pooler_output
                ids2 = self.tokenizer_2(prompt_2, padding='max_length', max_length=max_sequence_length, truncation=True,
                                        return_tensors='pt').input_ids
                if isinstance(self.text_encoder_2, T5Encoder):
                    prompt_embeds = self.text_encoder_2(ids2)
                else:
                    prompt_embeds = self.text_encoder_2(ids2.to(self.text_encoder_2.device))[0]
        prompt_embeds = prompt_embeds.to(device, torch.bfloat16).repeat_interleave(num_images_per_prompt, dim=0)
        pooled_prompt_embeds = pooled_prompt_embeds.to(device, torch.bfloat16).repeat_interleave(num_images_per_prompt, dim=0)
        return prompt_embeds, pooled_prompt_embeds

    def check_inputs(self, prompt, height, width, prompt_embeds, pooled_prompt_embeds, max_sequence_length):
        if height % (self.vae_scale_factor * 2) != 0 or width % (self.vae_scale_factor * 2) != 0:
            raise ValueError(f'`height` and `width` have to be divisible by {self.vae_scale_factor * 2}')
        if prompt is not None and prompt_embeds is not None:
            raise ValueError('Cannot forward both `prompt` and `prompt_embeds`.')
        if prompt is None and prompt_embeds is None:
            raise ValueError('Provide either `prompt` or `prompt_embeds`.')
        if prompt_embeds is not None and pooled_prompt_embeds is None:
            raise ValueError('If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed.')
        if max_sequence_length is not None and max_sequence_length > 512:
            raise ValueError(f'`max_sequence_length` cannot be greater than 512 but is {max_sequence_length}')

    @torch.inference_mode()
    def __call__(self, prompt: Union[str, List[str]] = None, prompt_2: Optional[Union[str, List[str]]] = None,
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 4,
                 total_substeps: int = 128, timestep_ratio: float = 0.5, temperature: Union[float, str] = 'auto',
                 guidance_scale: float = 3.5, num_images_per_prompt: Optional[int] = 1,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 latents: Optional[torch.FloatTensor] = None, prompt_embeds: Optional[torch.FloatTensor] = None,
                 pooled_prompt_embeds: Optional[torch.FloatTensor] = None, ip_adapter_image=None,
                 ip_adapter_image_embeds=None, output_type: Optional[str] = 'pil', return_dict: bool = True,
                 joint_attention_kwargs: Optional[Dict[str, Any]] = None,
                 callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ['latents'], max_sequence_length: int = 512):
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, prompt_embeds, pooled_prompt_embeds, max_sequence_length)
        if ip_adapter_image is not None or ip_adapter_image_embeds is not None:
            raise NotImplementedError('IP-Adapter inputs are outside the ArcFlow hot path')
        self._apply_lora_scale(float((joint_attention_kwargs or {}).get('scale', 1.0)))      # arcflux.py:147-154: scale_lora_layers around the forward
        if self.transformer is None or self.transformer.teacher_head:
            raise RuntimeError('load_arcflow_adapter() must be called before sampling (the plain FLUX head '
                               'predicts a single velocity, not an ArcFlow policy)')
        self._guidance_scale, self._interrupt = guidance_scale, False
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None:
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        device = self._execution_device
        prompt_embeds, pooled = self.encode_prompt(prompt, prompt_2, prompt_embeds, pooled_prompt_embeds, device,
                                                   num_images_per_prompt, max_sequence_length)
        B = batch_size * num_images_per_prompt
        latents, hp, wp = self._prepare_latents(B, height, width, generator, latents)
        guidance = torch.full((B,), guidance_scale, device=device, dtype=torch.float32) \
            if self.transformer.guidance_embeds else None

        def fwd(x, t, pe, prepared_step=None):
            return self.transformer(x, t, pe.to(device, torch.bfloat16), pooled, guidance, hp, wp, prepared_step=prepared_step)

        def prepare(sigmas):
            return self.transformer.prepare_steps(sigmas, pooled, guidance, B, hp * wp, prompt_embeds.shape[1])
        latents = self._denoise(latents, hp, wp, num_inference_steps, total_substeps, timestep_ratio, fwd,
                                callback_on_step_end, callback_on_step_end_tensor_inputs, prompt_embeds, prepare)
        if output_type == 'latent':
            image = latents
        else:
            if self.vae is None:
                raise RuntimeError("no VAE decoder attached: use output_type='latent' or set pipe.vae")
            from ..vae import AutoencoderKLDecoder
            if isinstance(self.vae, AutoencoderKLDecoder):      # HIP decoder: un-scaling + unpack fused into its first kernel
                image = self.vae.decode_packed(latents, hp, wp)
            else:                                               # any module with the diffusers AutoencoderKL interface
                lat = self._unpack(latents, hp, wp)
                lat = lat / self.vae.config.scaling_factor + self.vae.config.shift_factor
                image = self.vae.decode(lat.to(next(self.vae.parameters()).dtype), return_dict=False)[0]
            image = self._postprocess(image, output_type)
        self.maybe_free_model_hooks()
        if not return_dict:
            return (image,)
        return FluxPipelineOutput(images=image)
