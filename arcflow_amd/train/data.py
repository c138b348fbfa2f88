"""Data side of the distillation loop: the rank-strided, bucket-aware, resumable
sampler and the cached prompt-embedding items.

``DistributedSampler`` restates lakonlab/datasets/samplers/distributed_sampler.py:19-158 (golden-tested against the
reference class, tests/golden/g8_sampler.npz): every rank draws the SAME permutation from ``seed + epoch``, batches
are dealt round-robin to the ranks, a dataset exposing ``bucket_ids`` only yields single-bucket batches, and
``set_iter`` skips the batches a resumed run has already consumed.

``PromptEmbedCache`` reads the per-prompt files the reference's preprocessing writes
(lakonlab/datasets/image_prompts.py:286-309,357-437): a pickled dict with ``prompt``, ``prompt_embed_kwargs``
(``encoder_hidden_states`` + optional ``encoder_hidden_states_scale``, ``pooled_projections`` /
``encoder_hidden_states_mask``; or the legacy top-level ``prompt_embeds*`` spelling) and ``latent_size``; ``.zst`` files are read through ``zstd_io`` (the ``zstandard`` module, or pyarrow's zstd codec).
"""
from __future__ import annotations

import io
import math
import os
import pickle
from typing import Dict, Iterator, List, Optional, Sequence

import torch


class DistributedSampler:
    def __init__(self, dataset, num_replicas: int, rank: int, shuffle: bool = True, samples_per_gpu: int = 1, seed: int = 0):
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.shuffle, self.samples_per_gpu, self.seed = shuffle, samples_per_gpu, int(seed)
        self.epoch = 0
        self.skip_iter = 0
        self.update_sampler(dataset)

    def update_sampler(self, dataset, samples_per_gpu: Optional[int] = None):
        self.dataset = dataset
        if samples_per_gpu is not None:
            self.samples_per_gpu = samples_per_gpu
        spg, world = self.samples_per_gpu, self.num_replicas
        self.bucket_map = self.total_size_bucketwise = None
        if hasattr(dataset, 'bucket_ids'):
            bm: Dict[int, List[int]] = {}
            for i, b in enumerate(dataset.bucket_ids):
                bm.setdefault(b, []).append(i)
            self.bucket_map = dict(sorted(bm.items()))
            self.total_size_bucketwise = {}
            data_len = 0
            for b, idx in self.bucket_map.items():
                if len(idx) < spg:
                    raise ValueError('bucket smaller than samples_per_gpu: the sampler cannot pad it')
                self.total_size_bucketwise[b] = math.ceil(len(idx) / spg) * spg
                data_len += self.total_size_bucketwise[b]
        else:
            data_len = len(dataset)
        if data_len < world * spg:
            raise ValueError('dataset too small for num_replicas * samples_per_gpu')
        self.num_samples = math.ceil(data_len / world / spg) * spg
        self.total_size = self.num_samples * world

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def set_iter(self, iteration: int):
        self.skip_iter = iteration % (self.num_samples // self.samples_per_gpu)

    def __len__(self):
        return self.num_samples

    def __iter__(self) -> Iterator[int]:
        spg, world = self.samples_per_gpu, self.num_replicas
        g = None
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
        if self.bucket_map is None:
            idx = torch.randperm(len(self.dataset), generator=g).tolist() if g is not None else list(range(len(self.dataset)))
            idx += idx[:self.total_size - len(idx)]
            idx = idx[self.rank:self.total_size:world]
        else:
            rows = []
            for b, members in self.bucket_map.items():
                d = torch.tensor(members)
                if g is not None:
                    d = d[torch.randperm(len(d), generator=g)]
                pad = self.total_size_bucketwise[b] - d.numel()
                if pad:
                    d = torch.cat([d, d[:pad]])
                nb_total = self.total_size_bucketwise[b] // spg
                nb, left = nb_total // world, nb_total % world
                a = d[:nb * world * spg].reshape(nb, spg, world).permute(0, 2, 1).reshape(nb * world, spg)
                r = d[nb * world * spg:].reshape(spg, left).permute(1, 0)
                rows += [a, r]
            m = torch.cat(rows, dim=0)
            if g is not None:
                m = m[torch.randperm(m.size(0), generator=g)]
            nb_all = self.total_size // spg
            if nb_all > m.size(0):
                m = torch.cat([m, m[:nb_all - m.size(0)]], dim=0)
            idx = m[self.rank:nb_all:world].flatten().tolist()
        assert len(idx) == self.num_samples
        skip = self.skip_iter * spg
        assert skip < self.num_samples
        self.skip_iter = 0
        return iter(idx[skip:])


def _load_item(path: str):
    if path.endswith('.zst'):
        from . import zstd_io
        with open(path, 'rb') as f:                     # no silent fallback: zstd_io raises when no codec is present
            return pickle.load(io.BytesIO(zstd_io.decompress(f.read())))
    if path.endswith('.pt'):
        return torch.load(path, map_location='cpu', weights_only=False)
    with open(path, 'rb') as f:
        return pickle.load(f)


def _is_rank0() -> bool:
    try:
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
    except Exception:
        return int(os.environ.get('RANK', '0')) == 0


class PromptEmbedCache:
    """Items: ``{'ids', 'name', 'prompt_embed_kwargs': {...}, 'latent_size'}`` (image_prompts.py:357-383)."""
    LEGACY = {'prompt_embeds': 'encoder_hidden_states', 'prompt_embeds_scale': 'encoder_hidden_states_scale',
              'pooled_prompt_embeds': 'pooled_projections', 'prompt_embeds_mask': 'encoder_hidden_states_mask'}     # :86-91

    def __init__(self, cache_dir: str, datalist: Optional[Sequence[str]] = None, pad_seq_len: Optional[int] = None,
                 latent_size=(16, 128, 128), bucketize: bool = False, negative_prompt_embeds_path: Optional[str] = None,
                 size_index: Optional[str] = None):
        """negative_prompt_embeds_path: a torch.load-able dict of the NEGATIVE prompt's embeddings (same keys / legacy keys as an
        item), attached to every item as ``negative_prompt_embed_kwargs`` -- the reference's option of the same name
        (image_prompts.py:57-60,158-163,432), needed whenever the teacher runs true classifier-free guidance (Qwen config).
        size_index: JSON file {file name: [C, H, W]} with the latent size of every item.  ``bucketize`` needs the sizes up front;
        without an index it unpickles every cache file once (fine for thousands of items, hours for the reference's 3 M) and
        writes ``latent_sizes.json`` next to the cache so the next start is instant."""
        self.cache_dir, self.pad_seq_len, self.latent_size = cache_dir, pad_seq_len, tuple(latent_size)
        if datalist is None:
            datalist = sorted(f for f in os.listdir(cache_dir) if f.endswith(('.zst', '.pkl', '.pt')))
        self.files = list(datalist)
        self.negative_prompt_embed_kwargs = None
        if negative_prompt_embeds_path is not None:
            self.negative_prompt_embed_kwargs = self._parse(_load_item(negative_prompt_embeds_path), pad=False)
        if bucketize:          # one bucket per latent size: batches never mix resolutions
            index_path = size_index or os.path.join(cache_dir, 'latent_sizes.json')
            known = {}
            if os.path.exists(index_path):
                import json
                try:
                    with open(index_path) as f:
                        known = {k: tuple(v) for k, v in json.load(f).items()}
                except (ValueError, OSError, AttributeError, TypeError):
                    known = {}             # torn / foreign file (a killed job, another writer): rescan, rewrite below
            sizes, scanned = [], False
            for fn in self.files:
                if fn not in known:
                    known[fn] = tuple(_load_item(os.path.join(cache_dir, fn)).get('latent_size', self.latent_size))
                    scanned = True
                sizes.append(known[fn])
            if scanned and size_index is None and _is_rank0():
                # rank 0 only, temp file + os.replace: every rank of a data-parallel job constructs this dataset at the same moment, and
                # a reader must never see a half-written index
                try:
                    import json
                    tmp = f'{index_path}.tmp{os.getpid()}'
                    with open(tmp, 'w') as f:
                        json.dump({k: list(v) for k, v in known.items()}, f)
                    os.replace(tmp, index_path)
                except OSError:
                    pass               # read-only cache directory: scan again next time
            order = {s: i for i, s in enumerate(sorted(set(sizes)))}
            self.bucket_ids = [order[s] for s in sizes]
            self._index_sizes, self._index_path = sizes, index_path

    def __len__(self):
        return len(self.files)

    def _pad(self, x: torch.Tensor) -> torch.Tensor:
        if self.pad_seq_len is None or x.size(0) == self.pad_seq_len:
            return x
        if x.size(0) > self.pad_seq_len:
            return x[:self.pad_seq_len]
        return torch.cat([x, x.new_zeros((self.pad_seq_len - x.size(0),) + tuple(x.shape[1:]))], dim=0)

    def _parse(self, raw: dict, pad: bool = True) -> dict:
        """parse_prompt_embeds (image_prompts.py:286-309): legacy key names, the optional int8 scale, padding."""
        kw = dict(raw.get('prompt_embed_kwargs', {}))
        for old, new in self.LEGACY.items():
            if old in raw and new not in kw:
                kw[new] = raw[old]
        for new in self.LEGACY.values():
            if new in raw and new not in kw:
                kw[new] = raw[new]
        scale = kw.pop('encoder_hidden_states_scale', None)
        padf = self._pad if pad else (lambda x: x)
        if 'encoder_hidden_states' in kw:
            e = kw['encoder_hidden_states'].float()
            kw['encoder_hidden_states'] = padf(e * scale if scale is not None else e)
        if 'pooled_projections' in kw:
            kw['pooled_projections'] = kw['pooled_projections'].float()
        if 'encoder_hidden_states_mask' in kw:
            kw['encoder_hidden_states_mask'] = padf(kw['encoder_hidden_states_mask'])
        return kw

    def __getitem__(self, i: int) -> dict:
        raw = _load_item(os.path.join(self.cache_dir, self.files[i]))
        item = dict(ids=i, name=raw.get('prompt', ''), prompt_embed_kwargs=self._parse(raw),
                    latent_size=tuple(raw.get('latent_size', self.latent_size)))
        known = getattr(self, '_index_sizes', None)
        if known is not None and tuple(known[i]) != item['latent_size']:      # the index is never invalidated: a replaced item shows up here
            raise RuntimeError(f'{self.files[i]}: latent_size {item["latent_size"]} but the bucket index {self._index_path} says '
                               f'{tuple(known[i])} -- the cache changed after the index was written: delete the index and restart')
        if self.negative_prompt_embed_kwargs is not None:
            item['negative_prompt_embed_kwargs'] = self.negative_prompt_embed_kwargs
        return item


def collate(items: List[dict], device='cuda') -> dict:
    """A batch of cache items -> the ``cond`` dict ``ArcFlowDistiller.train_step`` takes."""
    sizes = {it['latent_size'] for it in items}
    if len(sizes) != 1:
        raise ValueError(f'mixed latent sizes in one batch: {sorted(sizes)} (use bucketize=True)')
    _, h, w = next(iter(sizes))
    def stack(kws):
        """[B, T, D] zero padded to the longest item; with masks (Qwen caches are pre-padded to a fixed length) the text is first
        truncated to the longest REAL length of the batch, as the reference does before the transformer
        (lakonlab/models/architecture/arcflow/arcqwen.py:325-330) -- zero pad tokens must not enter joint attention / the text RoPE."""
        embs = [k['encoder_hidden_states'] for k in kws]
        if all('encoder_hidden_states_mask' in k for k in kws):
            keep = max(int(k['encoder_hidden_states_mask'].sum()) for k in kws)
            embs = [e[:max(keep, 1)] for e in embs]
        T = max(e.size(0) for e in embs)
        return torch.stack([torch.cat([e, e.new_zeros(T - e.size(0), e.size(1))]) for e in embs]).to(device=device, dtype=torch.bfloat16)
    kws = [it['prompt_embed_kwargs'] for it in items]
    cond = dict(prompt_embeds=stack(kws), hp=h // 2, wp=w // 2)
    if 'pooled_projections' in kws[0]:
        cond['pooled'] = torch.stack([k['pooled_projections'] for k in kws]).to(device=device, dtype=torch.bfloat16)
    if 'negative_prompt_embed_kwargs' in items[0]:
        nk = [it['negative_prompt_embed_kwargs'] for it in items]
        cond['negative_prompt_embeds'] = stack(nk)
        if 'pooled_projections' in nk[0]:
            cond['negative_pooled'] = torch.stack([k['pooled_projections'] for k in nk]).to(device=device, dtype=torch.bfloat16)
    return cond
