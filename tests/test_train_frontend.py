"""Training front-end: sampler vs the reference class (golden g8), config reader, prompt cache."""
import os
import pickle

import numpy as np
import pytest
import torch

from arcflow_amd.train import config as CFG
from arcflow_amd.train import data as DATA

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'g8_sampler.npz')


class _DS:
    def __init__(self, n, buckets=None):
        self.n = n
        if buckets is not None:
            self.bucket_ids = buckets

    def __len__(self):
        return self.n


def test_sampler_matches_reference_golden():
    g = np.load(GOLD)
    n_cases = len([k for k in g.files if k.endswith('_meta')])
    assert n_cases == 7
    for ci in range(n_cases):
        n, world, spg, shuffle, seed, epoch, it, has_b = g[f'c{ci}_meta'].tolist()
        buckets = g[f'c{ci}_buckets'].tolist() if has_b == 1 else None
        seen = []
        for rank in range(world):
            s = DATA.DistributedSampler(_DS(n, buckets), world, rank, shuffle=bool(shuffle), samples_per_gpu=spg, seed=seed)
            s.set_epoch(epoch)
            s.set_iter(it)
            got = np.array(list(iter(s)), dtype=np.int64)
            assert np.array_equal(got, g[f'c{ci}_r{rank}']), (ci, rank)
            seen.append(got)
            if buckets is not None:          # every per-GPU batch comes from one bucket
                for b in got.reshape(-1, spg):
                    assert len({buckets[i] for i in b}) == 1
        assert len(list(iter(s))) == s.num_samples      # the skip applies once (resume), then full epochs


def test_sampler_rejects_tiny_datasets():
    with pytest.raises(ValueError):
        DATA.DistributedSampler(_DS(3), 2, 0, samples_per_gpu=4)
    with pytest.raises(ValueError):
        DATA.DistributedSampler(_DS(8, [0] * 7 + [1]), 2, 0, samples_per_gpu=2)


def test_config_reader_merges_bases_and_maps_to_distill_config(tmp_path):
    (tmp_path / '_base.py').write_text(
        "train_cfg = dict(diffusion_grad_clip=50.0, diffusion_grad_clip_begin_iter=100)\n"
        "optimizer = {'diffusion': dict(type='AdamW8bit', lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0,\n"
        "    paramwise_cfg=dict(custom_keys={'proj_out_loggamma': dict(lr_mult=0.1)}))}\n"
        "lr_config = dict(policy='fixed', warmup='linear', warmup_iters=100, warmup_ratio=0.001)\n"
        "runner = dict(type='R', ckpt_trainable_only=True, ckpt_fp16=True, ckpt_fp16_ema=True)\n"
        "model = dict(diffusion=dict(denoising=dict(freeze_exclude_autocast_dtype='bfloat16')))\n")
    (tmp_path / 'exp.py').write_text(
        "_base_ = ['./_base.py']\nk = 16\nname = f'exp_k{k}'\n"
        "model = dict(type='LatentDiffusionTextImage', diffusion=dict(type='ArcFlowImitationDataFree', policy_type='ArcFlow',\n"
        "  denoising=dict(type='ArcFluxTransformer2DModel', num_gaussians=k, logweights_channels=4, in_channels=64, num_layers=19,\n"
        "    num_single_layers=38, attention_head_dim=128, num_attention_heads=24, joint_attention_dim=4096,\n"
        "    pooled_projection_dim=768, guidance_embeds=True, use_lora=True, lora_dropout=0.05, lora_rank=256),\n"
        "  flow_loss=dict(type='DiffusionMSELoss', rescale_cfg=dict(scale=30.0)),\n"
        "  timestep_sampler=dict(type='ContinuousTimeStepSampler', shift=3.2)))\n"
        "train_cfg = dict(num_decay_iters=2000, window_substeps=3, gm_dropout=0.1, num_intermediate_states=4,\n"
        "  distilled_guidance_scale=3.5, nfe=2, timestep_ratio=1.0, total_substeps=128)\n"
        "data = dict(train_dataloader=dict(samples_per_gpu=4))\n"
        "checkpoint_config = dict(interval=500, out_dir='checkpoints/')\ntotal_iters = 10000\n"
        "custom_hooks = [dict(type='ExponentialMovingAverageHookMod', start_iter=100, momentum_cfg=dict(gamma=7.0))]\n"
        "resume_from = f'checkpoints/{name}/latest.pth'\n")
    cfg = CFG.load_config(str(tmp_path / 'exp.py'))
    assert cfg['model']['diffusion']['denoising']['freeze_exclude_autocast_dtype'] == 'bfloat16'      # from the base
    assert cfg['train_cfg']['diffusion_grad_clip'] == 50.0 and cfg['train_cfg']['nfe'] == 2           # dicts merge
    fam, eng, dc, run = CFG.distill_setup(cfg)
    assert fam == 'flux' and eng['num_double'] == 19 and eng['num_single'] == 38 and eng['joint_dim'] == 4096
    assert (dc.lr, dc.betas, dc.loggamma_lr_mult, dc.warmup_iters, dc.warmup_ratio) == (1e-4, (0.9, 0.95), 0.1, 100, 0.001)
    assert (dc.grad_clip, dc.grad_clip_begin_iter, dc.loss_scale, dc.shift, dc.lora_rank, dc.lora_dropout) == (50.0, 100, 30.0, 3.2, 256, 0.05)
    assert (dc.ema_gamma, dc.ema_start_iter, dc.gm_dropout, dc.num_decay_iters) == (7.0, 100, 0.1, 2000)
    assert run['samples_per_gpu'] == 4 and run['ckpt_fp16'] and run['ckpt_dir'] == 'checkpoints/exp_k16'
    assert run['resume_from'] == 'checkpoints/exp_k16/latest.pth' and run['lora_dropout'] == 0.05
    cfg2 = CFG.apply_options(cfg, {'train_cfg.nfe': 4, 'total_iters': 10})
    assert cfg2['train_cfg']['nfe'] == 4 and cfg2['train_cfg']['gm_dropout'] == 0.1 and cfg2['total_iters'] == 10
    # the fp8 forward options (BASELINE.json configs[4]; no key of the reference's configs): off by default, --cfg-options turns them on
    assert not dc.teacher_fp8 and not dc.student_fp8
    dc3 = CFG.distill_setup(CFG.apply_options(cfg, {'train_cfg.teacher_fp8': True, 'train_cfg.student_fp8': True}))[2]
    assert dc3.teacher_fp8 and dc3.student_fp8


@pytest.mark.skipif(not os.path.isdir('/root/reference/configs'), reason='reference tree not present')
def test_reference_configs_load():
    for path, fam, nd, tgs, decay in (('/root/reference/configs/flux/arcflux_2nfe_k16.py', 'flux', 19, 1.0, 2000),
                                       ('/root/reference/configs/qwen/arcqwen_2nfe_k16.py', 'qwen', 60, 4.0, 1000)):
        f, eng, dc, run = CFG.distill_setup(CFG.load_config(path))
        assert (f, eng['num_double'], dc.teacher_guidance_scale, dc.num_decay_iters) == (fam, nd, tgs, decay)
        assert dc.lora_rank == 256 and dc.lora_dropout == 0.05 and dc.lr == 1e-4 and dc.grad_clip == 50.0 and dc.loss_scale == 30.0 and dc.nfe == 2


def test_prompt_cache_items_and_collate(tmp_path):
    g = torch.Generator().manual_seed(0)
    for i, (T, size) in enumerate([(5, (16, 8, 8)), (7, (16, 8, 8)), (4, (16, 4, 12))]):
        e, p = torch.randn(T, 32, generator=g).half(), torch.randn(16, generator=g).half()
        if i == 1:      # legacy spelling (image_prompts.py:86-91)
            item = dict(prompt=f'p{i}', prompt_embeds=e, prompt_embeds_scale=2.0, pooled_prompt_embeds=p, latent_size=size)
        else:
            item = dict(prompt=f'p{i}', prompt_embed_kwargs=dict(encoder_hidden_states=e, encoder_hidden_states_scale=2.0,
                                                                  pooled_projections=p), latent_size=size)
        with open(tmp_path / f'{i:04d}.pkl', 'wb') as f:
            pickle.dump(item, f)
    ds = DATA.PromptEmbedCache(str(tmp_path), pad_seq_len=6, bucketize=True)
    assert len(ds) == 3 and ds.bucket_ids == [1, 1, 0]
    a, b = ds[0], ds[1]
    assert a['prompt_embed_kwargs']['encoder_hidden_states'].shape == (6, 32) and a['name'] == 'p0'
    assert torch.all(a['prompt_embed_kwargs']['encoder_hidden_states'][5] == 0)                 # zero padding
    assert b['prompt_embed_kwargs']['encoder_hidden_states'].shape == (6, 32)                   # truncation
    raw = pickle.load(open(tmp_path / '0000.pkl', 'rb'))
    assert torch.allclose(a['prompt_embed_kwargs']['encoder_hidden_states'][:5], raw['prompt_embed_kwargs']['encoder_hidden_states'].float() * 2.0)
    cond = DATA.collate([a, b], device='cpu')
    assert cond['prompt_embeds'].shape == (2, 6, 32) and cond['pooled'].shape == (2, 16) and (cond['hp'], cond['wp']) == (4, 4)
    with pytest.raises(ValueError):
        DATA.collate([a, ds[2]], device='cpu')


def test_prompt_cache_reads_and_writes_zstd_items(tmp_path):
    """The reference's cache items are zstd-compressed pickles (image_prompts.py:357-383: `<stem>.zst`).  This image has no
    `zstandard` module; `zstd_io` goes through pyarrow's zstd codec.  Checked: a standard frame written here reads back, a
    frame with an UNKNOWN content size (what a streaming writer such as the reference's produces) reads back, the dataset
    loads `.zst` files, and `write_cache` emits them."""
    from arcflow_amd.train import zstd_io
    assert zstd_io.available()
    blob = b'prompt cache ' * 1000
    assert zstd_io.decompress(zstd_io.compress(blob)) == blob
    # zstd frame of the 3 bytes b'abc' without a content-size field: magic, frame header 0x00 0x58 (single segment off, window
    # descriptor), one raw last block (header 0x19 0x00 0x00 = last, raw, size 3) -- hand-assembled from RFC 8878
    frame = bytes.fromhex('28b52ffd' '00' '58' '190000') + b'abc'
    assert zstd_io.decompress(frame) == b'abc'
    assert zstd_io.decompress(frame + zstd_io.compress(b'def')) == b'abcdef'          # concatenated frames
    g = torch.Generator().manual_seed(3)
    e, p = torch.randn(5, 32, generator=g).half(), torch.randn(16, generator=g).half()
    item = dict(prompt='a cat', prompt_embed_kwargs=dict(encoder_hidden_states=e, pooled_projections=p), latent_size=(16, 8, 8))
    with open(tmp_path / '0000.zst', 'wb') as f:
        f.write(zstd_io.compress(pickle.dumps(item)))
    ds = DATA.PromptEmbedCache(str(tmp_path), pad_seq_len=6)
    assert len(ds) == 1 and ds[0]['name'] == 'a cat'
    assert torch.equal(ds[0]['prompt_embed_kwargs']['encoder_hidden_states'][:5], e.float())

    class Enc:          # the PromptEncoder surface write_cache uses
        def encode(self, prompts):
            n = len(prompts)
            return dict(encoder_hidden_states=torch.ones(n, 4, 8), pooled_projections=torch.ones(n, 3))

    from arcflow_amd.train.prompts import write_cache
    out = tmp_path / 'written'
    names = write_cache(Enc(), ['x', 'y', 'z'], str(out), latent_size=(16, 8, 8), batch=2)
    assert names == ['00000000', '00000001', '00000002'] and sorted(os.listdir(out)) == [n + '.zst' for n in names]
    ds2 = DATA.PromptEmbedCache(str(out))
    assert len(ds2) == 3 and ds2[2]['name'] == 'z' and ds2[2]['prompt_embed_kwargs']['encoder_hidden_states'].shape == (4, 8)


def test_prompt_cache_negative_embeds_mask_truncation_and_size_index(tmp_path):
    """(medium) a --data-dir cache must deliver negative_prompt_embeds for the true-CFG (Qwen) teacher; (low) masks
    truncate the text to the longest real length of the batch (arcqwen.py:325-330); (low) the bucket sizes come from an index."""
    import json
    g = torch.Generator().manual_seed(1)
    cache = tmp_path / 'cache'
    cache.mkdir()
    for i, real in enumerate([3, 5]):
        e = torch.zeros(8, 16)
        e[:real] = torch.randn(real, 16, generator=g)
        m = torch.zeros(8, dtype=torch.long)
        m[:real] = 1
        pickle.dump(dict(prompt=f'q{i}', prompt_embed_kwargs=dict(encoder_hidden_states=e, encoder_hidden_states_mask=m),
                         latent_size=(16, 8, 8)), open(cache / f'{i:03d}.pkl', 'wb'))
    neg = dict(prompt_embeds=torch.randn(2, 16, generator=g), prompt_embeds_mask=torch.ones(2, dtype=torch.long))   # legacy keys
    torch.save(neg, tmp_path / 'neg.pt')
    ds = DATA.PromptEmbedCache(str(cache), bucketize=True, negative_prompt_embeds_path=str(tmp_path / 'neg.pt'))
    assert json.load(open(cache / 'latent_sizes.json')) == {'000.pkl': [16, 8, 8], '001.pkl': [16, 8, 8]}     # written once
    (cache / '000.pkl').rename(cache / '000.pkl.moved')          # a second start must not unpickle the items for their sizes
    ds2 = DATA.PromptEmbedCache(str(cache), datalist=['000.pkl', '001.pkl'], bucketize=True)
    assert ds2.bucket_ids == [0, 0]
    (cache / '000.pkl.moved').rename(cache / '000.pkl')
    cond = DATA.collate([ds[0], ds[1]], device='cpu')
    assert cond['prompt_embeds'].shape == (2, 5, 16)                                  # 8 padded tokens -> 5 real ones
    assert torch.all(cond['prompt_embeds'][0, 3:] == 0)
    assert cond['negative_prompt_embeds'].shape == (2, 2, 16)
    assert torch.allclose(cond['negative_prompt_embeds'][1].float(), neg['prompt_embeds'].bfloat16().float())


_TINY_CFG = """
name = 'tiny'
model = dict(diffusion=dict(type='ArcFlowImitationDataFree', policy_type='ArcFlow', policy_kwargs=dict(),
    denoising=dict(type='ArcFluxTransformer2DModel', num_gaussians=16, logweights_channels=4, in_channels=64, num_layers=1,
        num_single_layers=1, attention_head_dim=128, num_attention_heads=2, joint_attention_dim=128, pooled_projection_dim=64,
        guidance_embeds=True, use_lora=True, lora_rank=64, lora_dropout=0.05),
    flow_loss=dict(type='DiffusionMSELoss', rescale_cfg=dict(scale=30.0)), timestep_sampler=dict(shift=3.2)))
train_cfg = dict(num_decay_iters=4, window_substeps=3, gm_dropout=0.1, num_intermediate_states=4, nfe=2, timestep_ratio=1.0,
                 total_substeps=128, diffusion_grad_clip=50.0, diffusion_grad_clip_begin_iter=1)
optimizer = {'diffusion': dict(type='AdamW8bit', lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0)}
lr_config = dict(warmup_iters=2, warmup_ratio=0.001)
runner = dict(ckpt_fp16=True, ckpt_fp16_ema=True)
data = dict(train_dataloader=dict(samples_per_gpu=2))
checkpoint_config = dict(interval=2, out_dir='checkpoints/')
total_iters = 4
custom_hooks = [dict(type='ExponentialMovingAverageHookMod', start_iter=1, momentum_cfg=dict(gamma=7.0))]
"""


@pytest.mark.gpu
def test_train_cli_runs_saves_and_resumes(tmp_path):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfgp = tmp_path / 'tiny.py'
    cfgp.write_text(_TINY_CFG)
    base = [sys.executable, os.path.join(root, 'tools', 'train.py'), str(cfgp), '--synthetic', '--work-dir', str(tmp_path / 'w'),
            '--latent-tokens', '8', '8', '--diff_seed']
    r = subprocess.run(base + ['--iters', '3'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    logs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    assert [l['iter'] for l in logs] == [1, 2, 3] and all(l['loss'] == l['loss'] for l in logs)
    ck = tmp_path / 'w' / 'checkpoints'
    assert (ck / 'iter_2.pth').exists() and (ck / 'iter_3.pth').exists() and os.readlink(ck / 'latest.pth') == 'iter_3.pth'
    sd = torch.load(ck / 'iter_3.pth', map_location='cpu', weights_only=False)['state_dict']
    assert sd['diffusion.denoising.proj_out_means.weight'].dtype == torch.float16          # ckpt_fp16
    r2 = subprocess.run(base + ['--export', str(tmp_path / 'adapter')], capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    logs2 = [json.loads(l) for l in r2.stdout.splitlines() if l.startswith('{')]
    assert [l['iter'] for l in logs2] == [4] and 'resumed from' in r2.stdout
    assert (tmp_path / 'adapter' / 'config.json').exists() and (tmp_path / 'adapter' / 'diffusion_pytorch_model.safetensors').exists()


def test_t5_relative_bucket_function_matches_transformers():
    """CPU: the restated T5 bucket function (host side of the bias table) against transformers' own static method."""
    from transformers.models.t5.modeling_t5 import T5Attention
    from arcflow_amd.text_encoders import t5_relative_buckets
    d = torch.arange(-600, 601)
    for nb, md in ((32, 128), (16, 64), (64, 256)):
        assert torch.equal(t5_relative_buckets(d, nb, md), T5Attention._relative_position_bucket(d, True, nb, md))
