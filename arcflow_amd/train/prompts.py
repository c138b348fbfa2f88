"""Prompt -> conditioning for the distillation loop: the two data modes of the reference (configs/flux/README.md "Data
Preparation"): (a) pre-encode the prompts once into the cache the dataset reads (its ``cache_image_prompt_data.py`` step), or
(b) keep the text encoder in the training process (``text_encoder=dict(type='PretrainedFluxTextEncoder')`` in
``_ddp_train.py``; lakonlab/models/architecture/diffusers/pretrained.py:152-238).  Both run the HIP prompt encoders of
``arcflow_amd/text_encoders.py``; tokenisation stays with the transformers tokenizers of the model snapshot.
"""
from __future__ import annotations

import io
import json
import os
import pickle
from typing import Dict, Iterable, List, Optional, Sequence

import torch


class PromptEncoder:
    """``encode(prompts) -> prompt_embed_kwargs`` with the reference wrappers' keys: ``encoder_hidden_states`` (+
    ``pooled_projections`` for FLUX, ``encoder_hidden_states_mask`` for Qwen-Image)."""

    def __init__(self, family: str, pipe, max_sequence_length: int = 512, pad_seq_len: Optional[int] = None):
        assert family in ('flux', 'qwen')
        if pad_seq_len is not None:
            assert pad_seq_len >= max_sequence_length
        self.family, self.pipe, self.max_len, self.pad = family, pipe, max_sequence_length, pad_seq_len

    @classmethod
    def from_snapshot(cls, family: str, root: str, **kw) -> 'PromptEncoder':
        """root: a local FLUX.1-dev / Qwen-Image snapshot with ``text_encoder*/`` and ``tokenizer*/`` (no network here)."""
        from transformers import AutoTokenizer
        from .. import text_encoders as TE
        from ..pipelines import ArcFluxPipeline, ArcQwenImagePipeline
        if family == 'flux':
            pipe = ArcFluxPipeline()
            pipe.text_encoder = TE.load_clip_text_encoder(os.path.join(root, 'text_encoder'))
            pipe.text_encoder_2 = TE.load_t5_encoder(os.path.join(root, 'text_encoder_2'))
            pipe.tokenizer = AutoTokenizer.from_pretrained(os.path.join(root, 'tokenizer'))
            pipe.tokenizer_2 = AutoTokenizer.from_pretrained(os.path.join(root, 'tokenizer_2'))
        else:
            pipe = ArcQwenImagePipeline()
            pipe.text_encoder = TE.load_qwen25_text_encoder(os.path.join(root, 'text_encoder'))
            pipe.tokenizer = AutoTokenizer.from_pretrained(os.path.join(root, 'tokenizer'))
        return cls(family, pipe, **kw)

    @torch.no_grad()
    def encode(self, prompts: Sequence[str]) -> Dict[str, torch.Tensor]:
        prompts = list(prompts)
        if self.family == 'flux':
            pe, pooled = self.pipe.encode_prompt(prompts, None, None, None, 'cuda', 1, self.max_len)
            return dict(encoder_hidden_states=pe, pooled_projections=pooled)
        pe, mask = self.pipe.encode_prompt(prompts, max_sequence_length=self.max_len)
        if self.pad is not None and pe.shape[1] < self.pad:          # PretrainedQwenImageTextEncoder.forward pad_seq_len
            pe = torch.nn.functional.pad(pe, (0, 0, 0, self.pad - pe.shape[1]))
            mask = torch.nn.functional.pad(mask, (0, self.pad - mask.shape[1]))
        return dict(encoder_hidden_states=pe, encoder_hidden_states_mask=mask)

    def cond(self, prompts: Sequence[str], hp: int, wp: int, negative_prompts: Optional[Sequence[str]] = None) -> dict:
        """The ``cond`` dict of ``ArcFlowDistiller.train_step`` for a batch of prompts."""
        e = self.encode(prompts)
        pe = e['encoder_hidden_states']
        if 'encoder_hidden_states_mask' in e:
            pe = pe[:, :int(e['encoder_hidden_states_mask'].sum(1).max())]
        c = dict(prompt_embeds=pe.to(torch.bfloat16), hp=hp, wp=wp)
        if 'pooled_projections' in e:
            c['pooled'] = e['pooled_projections'].to(torch.bfloat16)
        if negative_prompts is not None:
            n = self.encode(negative_prompts)
            c['negative_prompt_embeds'] = n['encoder_hidden_states'].to(torch.bfloat16)
            if 'pooled_projections' in n:
                c['negative_pooled'] = n['pooled_projections'].to(torch.bfloat16)
        return c


def write_cache(encoder: PromptEncoder, prompts: Iterable[str], out_dir: str, latent_size=(16, 128, 128), batch: int = 8,
                start_index: int = 0, compress: Optional[bool] = None) -> List[str]:
    """Encode ``prompts`` and write one item per prompt in the layout ``PromptEmbedCache`` / the reference's ``ImagePrompt``
    dataset read (image_prompts.py:357-383): a pickled dict {prompt, prompt_embed_kwargs (fp16, unpadded), latent_size}.
    Files are ``<index>.zst`` (what the reference reads) when a zstd codec is present (``zstd_io``), else ``<index>.pkl``.
    Returns the datalist (file stems); also written to ``<out_dir>.jsonl``."""
    from . import zstd_io
    if compress is None:
        compress = zstd_io.available()
    if compress and not zstd_io.available():
        raise RuntimeError('compress=True needs the zstandard module or pyarrow with the zstd codec')
    os.makedirs(out_dir, exist_ok=True)
    names: List[str] = []
    buf: List[str] = []

    def flush():
        if not buf:
            return
        e = encoder.encode(buf)
        for i, p in enumerate(buf):
            kw = {}
            hs = e['encoder_hidden_states'][i]
            if 'encoder_hidden_states_mask' in e:
                m = e['encoder_hidden_states_mask'][i].bool()
                hs = hs[m]
                kw['encoder_hidden_states_mask'] = torch.ones(int(m.sum()), dtype=torch.long)
            kw['encoder_hidden_states'] = hs.to('cpu', torch.float16)
            if 'pooled_projections' in e:
                kw['pooled_projections'] = e['pooled_projections'][i].to('cpu', torch.float16)
            item = dict(prompt=p, prompt_embed_kwargs=kw, latent_size=tuple(latent_size))
            stem = f'{start_index + len(names):08d}'
            raw = pickle.dumps(item, protocol=pickle.HIGHEST_PROTOCOL)
            if compress:
                with open(os.path.join(out_dir, stem + '.zst'), 'wb') as f:
                    f.write(zstd_io.compress(raw, level=3))
            else:
                with open(os.path.join(out_dir, stem + '.pkl'), 'wb') as f:
                    f.write(raw)
            names.append(stem)
        buf.clear()

    for p in prompts:
        buf.append(p)
        if len(buf) == batch:
            flush()
    flush()
    with io.open(out_dir.rstrip('/') + '.jsonl', 'w', encoding='utf-8') as f:
        for n in names:
            f.write(json.dumps(n) + '\n')
    return names
