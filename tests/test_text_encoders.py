"""Prompt encoders (arcflow_amd/text_encoders.py) against the real transformers modules (fp32, CPU, random-init small
configs of the same architecture): T5 v1.1 encoder, CLIP text model, Qwen2.5-VL language model."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()


def _bf16_weights(m):
    # the engine holds bf16 weights: give the oracle the same (rounded) values
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.bfloat16().float())
    return m


@pytest.mark.parametrize('S', [64, 77, 200])
def test_attention_ext_vs_torch(S):
    import ctypes as C
    from arcflow_amd import _lib
    from arcflow_amd.text_encoders import _p, _s
    lib = _lib.load()
    g = torch.Generator().manual_seed(S)
    for (H, Hkv, d, causal, use_bias) in [(4, 4, 64, False, True), (3, 3, 64, True, False), (8, 2, 128, True, False), (2, 2, 128, False, False)]:
        Dq, Dk = H * d, Hkv * d
        qkv = (torch.randn(S, Dq + 2 * Dk, generator=g) * 0.7).bfloat16()
        scale = 1.0 if use_bias else d ** -0.5
        bias = (torch.randn(H, 2 * S - 1, generator=g)) if use_bias else None
        q = qkv[:, :Dq].float().view(S, H, d).transpose(0, 1)
        k = qkv[:, Dq:Dq + Dk].float().view(S, Hkv, d).transpose(0, 1).repeat_interleave(H // Hkv, 0)
        v = qkv[:, Dq + Dk:].float().view(S, Hkv, d).transpose(0, 1).repeat_interleave(H // Hkv, 0)
        s = q @ k.transpose(1, 2) * scale
        idx = torch.arange(S)
        if use_bias:
            s = s + bias[:, (idx[None, :] - idx[:, None]) + S - 1]
        if causal:
            s = s.masked_fill(idx[None, :] > idx[:, None], float('-inf'))
        ref = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(S, Dq)
        qg = qkv.cuda()
        o = torch.empty(S, Dq, dtype=torch.bfloat16, device='cuda')
        ws = torch.empty(lib.afx_attention_ext_ws_bytes(1, Hkv, S, d), dtype=torch.uint8, device='cuda')
        ld = qg.stride(0)
        _lib.check(lib.afx_attention_ext_bf16(_p(qg), ld, _p(qg[:, Dq:]), ld, _p(qg[:, Dq + Dk:]), ld, _p(o), Dq, _p(ws), 1, H, Hkv, S, d,
                                              scale, int(causal), _p(None if bias is None else bias.cuda()), _s()))
        assert _rel(o, ref) < 1.5e-2, (H, Hkv, d, causal, use_bias, _rel(o, ref))


def test_t5_encoder_vs_transformers():
    from transformers import T5Config, T5EncoderModel
    from arcflow_amd.text_encoders import T5Encoder, t5_relative_buckets
    torch.manual_seed(0)
    cfg = T5Config(vocab_size=300, d_model=128, d_kv=64, d_ff=256, num_layers=3, num_heads=4, feed_forward_proj='gated-gelu',
                   relative_attention_num_buckets=32, relative_attention_max_distance=128, dropout_rate=0.0)
    m = _bf16_weights(T5EncoderModel(cfg).eval())
    with torch.no_grad():      # default init leaves the bias table and norms trivial: make them matter
        for n, p in m.named_parameters():
            if 'relative_attention_bias' in n:
                p.copy_((torch.randn_like(p) * 0.5).bfloat16().float())
            if 'layer_norm' in n:
                p.copy_((1 + 0.2 * torch.randn_like(p)).bfloat16().float())
    # the bucket function against the module's own
    from transformers.models.t5.modeling_t5 import T5Attention
    d = torch.arange(-300, 301)
    assert torch.equal(t5_relative_buckets(d), T5Attention._relative_position_bucket(d, True, 32, 128))
    ids = torch.randint(0, 300, (2, 96))
    ids[1, 40:] = 0                                        # padding tokens are ordinary tokens (no mask is passed)
    with torch.no_grad():
        ref = m(input_ids=ids).last_hidden_state
    enc = T5Encoder(m.state_dict(), num_layers=3, num_heads=4, d_kv=64)
    out = enc(ids)
    assert out.shape == ref.shape and _rel(out, ref) < 2e-2, _rel(out, ref)


def test_clip_text_encoder_vs_transformers():
    from transformers import CLIPTextConfig, CLIPTextModel
    from arcflow_amd.text_encoders import CLIPTextEncoder
    torch.manual_seed(1)
    for eos_cfg in (2, 299):
        cfg = CLIPTextConfig(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                             max_position_embeddings=77, hidden_act='quick_gelu', eos_token_id=eos_cfg, bos_token_id=298, pad_token_id=299)
        m = _bf16_weights(CLIPTextModel(cfg).eval())
        ids = torch.randint(0, 290, (2, 77))
        ids[:, 0] = 298
        ids[0, 20:] = 299                                  # eos (= highest id) then padding with the same id
        ids[1, 50:] = 299
        with torch.no_grad():
            ref = m(input_ids=ids)
        enc = CLIPTextEncoder(m.state_dict(), num_layers=3, num_heads=2, eos_token_id=eos_cfg)
        hs, pooled = enc(ids)
        assert _rel(hs, ref.last_hidden_state) < 2e-2 and _rel(pooled, ref.pooler_output) < 2e-2


def test_qwen25_text_encoder_vs_transformers():
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    from arcflow_amd.text_encoders import Qwen25TextEncoder
    torch.manual_seed(2)
    text = dict(vocab_size=400, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2, num_key_value_heads=1,
                max_position_embeddings=512, rope_theta=1e6, rms_norm_eps=1e-6, tie_word_embeddings=False,
                rope_scaling=dict(type='mrope', mrope_section=[16, 24, 24]))
    vis = dict(depth=1, hidden_size=64, intermediate_size=128, num_heads=2, out_hidden_size=256)
    try:
        cfg = Qwen2_5_VLConfig(text_config=text, vision_config=vis)
    except TypeError:
        cfg = Qwen2_5_VLConfig(vision_config=vis, **text)
    m = _bf16_weights(Qwen2_5_VLForConditionalGeneration(cfg).eval())
    with torch.no_grad():
        for n, p in m.named_parameters():
            if 'layernorm' in n or n.endswith('norm.weight'):
                p.copy_((1 + 0.2 * torch.randn_like(p)).bfloat16().float())
            if n.endswith('proj.bias'):
                p.copy_((0.3 * torch.randn_like(p)).bfloat16().float())
    ids = torch.randint(0, 390, (2, 90))
    mask = torch.ones(2, 90, dtype=torch.long)
    mask[1, 70:] = 0
    with torch.no_grad():
        ref = m(input_ids=ids, attention_mask=mask, output_hidden_states=True).hidden_states[-1]
    enc = Qwen25TextEncoder(m.state_dict(), num_layers=3, num_heads=2, num_kv_heads=1)
    out = enc(ids, mask)
    assert _rel(out[0], ref[0]) < 2e-2, _rel(out[0], ref[0])
    assert _rel(out[1, :70], ref[1, :70]) < 2e-2


class _Tok:
    """Stand-in tokenizer (the real ones are transformers tokenizers read from the model snapshot): hashes characters to ids."""

    def __init__(self, vocab, pad_id, eos_id=None, bos_id=None, left_pad=False):
        self.vocab, self.pad, self.eos, self.bos, self.left = vocab, pad_id, eos_id, bos_id, left_pad

    def __call__(self, texts, padding=True, max_length=None, truncation=True, return_tensors='pt'):
        rows = []
        for t in texts:
            ids = [(ord(c) * 7) % (self.vocab - 10) for c in t]
            ids = ([self.bos] if self.bos is not None else []) + ids + ([self.eos] if self.eos is not None else [])
            rows.append(ids[:max_length])
        L = max_length if padding == 'max_length' else max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad, dtype=torch.long)
        mask = torch.zeros(len(rows), L, dtype=torch.long)
        for i, r in enumerate(rows):
            sl = slice(L - len(r), L) if self.left else slice(0, len(r))
            ids[i, sl] = torch.tensor(r)
            mask[i, sl] = 1
        return type('Enc', (), dict(input_ids=ids, attention_mask=mask))()


def test_qwen_encode_prompt_matches_transformers_path():
    """pipe.encode_prompt (template, valid-token extraction, 34-token drop, zero padding) with the HIP encoder vs the same
    host logic around the transformers module; a left-padding tokenizer checks the mask handling."""
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    from arcflow_amd.pipelines import ArcQwenImagePipeline
    from arcflow_amd.text_encoders import Qwen25TextEncoder
    torch.manual_seed(3)
    text = dict(vocab_size=400, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                max_position_embeddings=2048, rope_theta=1e6, rms_norm_eps=1e-6, tie_word_embeddings=False,
                rope_scaling=dict(type='mrope', mrope_section=[16, 24, 24]))
    cfg = Qwen2_5_VLConfig(text_config=text, vision_config=dict(depth=1, hidden_size=64, intermediate_size=128, num_heads=2, out_hidden_size=256))
    m = _bf16_weights(Qwen2_5_VLForConditionalGeneration(cfg).eval())
    prompts = ['a red cube on a glass table', 'two cats']
    for left in (False, True):
        pipe = ArcQwenImagePipeline()
        pipe.tokenizer = _Tok(400, pad_id=399, left_pad=left)
        pipe.text_encoder = Qwen25TextEncoder(m.state_dict(), num_layers=2, num_heads=2, num_kv_heads=1)
        emb, mask = pipe.encode_prompt(prompts)
        ref_pipe = ArcQwenImagePipeline()
        ref_pipe.tokenizer, ref_pipe.text_encoder = pipe.tokenizer, m
        with torch.no_grad():
            remb, rmask = ref_pipe.encode_prompt(prompts)
        assert emb.shape == remb.shape and torch.equal(mask.cpu(), rmask.cpu())
        assert mask.sum(1).tolist() == [len(pipe.prompt_template_encode.format(p)) - 34 for p in prompts]
        assert _rel(emb, remb) < 2e-2, (left, _rel(emb, remb))


def test_flux_encode_prompt_with_hip_encoders():
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel
    from arcflow_amd.pipelines import ArcFluxPipeline
    from arcflow_amd.text_encoders import CLIPTextEncoder, T5Encoder
    torch.manual_seed(4)
    t5 = _bf16_weights(T5EncoderModel(T5Config(vocab_size=300, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2,
                                               feed_forward_proj='gated-gelu', dropout_rate=0.0)).eval())
    clip = _bf16_weights(CLIPTextModel(CLIPTextConfig(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                                                      max_position_embeddings=77, eos_token_id=2, bos_token_id=298, pad_token_id=299)).eval())
    pipe = ArcFluxPipeline()
    pipe.tokenizer, pipe.tokenizer_2 = _Tok(300, pad_id=299, eos_id=299, bos_id=298), _Tok(300, pad_id=0, eos_id=1)
    pipe.text_encoder = CLIPTextEncoder(clip.state_dict(), num_layers=2, num_heads=1, eos_token_id=2)
    pipe.text_encoder_2 = T5Encoder(t5.state_dict(), num_layers=2, num_heads=2, d_kv=64)
    pe, pooled = pipe.encode_prompt('a photo of a fox', None, None, None, 'cuda', 2, 128)
    with torch.no_grad():
        ids = pipe.tokenizer(['a photo of a fox'], padding='max_length', max_length=77).input_ids
        ids2 = pipe.tokenizer_2(['a photo of a fox'], padding='max_length', max_length=128).input_ids
        rp, re = clip(input_ids=ids).pooler_output, t5(input_ids=ids2).last_hidden_state
    assert pe.shape == (2, 128, 128) and pooled.shape == (2, 64)
    assert _rel(pe[0], re[0]) < 2e-2 and _rel(pooled[1], rp[0]) < 2e-2


def test_prompt_cache_round_trip_and_online_cond(tmp_path):
    """tools/cache_prompts.py path: prompts -> HIP encoders -> cache files -> PromptEmbedCache -> collate == online cond."""
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel
    from arcflow_amd.pipelines import ArcFluxPipeline
    from arcflow_amd.text_encoders import CLIPTextEncoder, T5Encoder
    from arcflow_amd.train import data as DATA
    from arcflow_amd.train.prompts import PromptEncoder, write_cache
    torch.manual_seed(5)
    t5 = _bf16_weights(T5EncoderModel(T5Config(vocab_size=300, d_model=128, d_kv=64, d_ff=256, num_layers=1, num_heads=2,
                                               feed_forward_proj='gated-gelu', dropout_rate=0.0)).eval())
    clip = _bf16_weights(CLIPTextModel(CLIPTextConfig(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1,
                                                      max_position_embeddings=77, eos_token_id=2, bos_token_id=298, pad_token_id=299)).eval())
    pipe = ArcFluxPipeline()
    pipe.tokenizer, pipe.tokenizer_2 = _Tok(300, pad_id=299, eos_id=299, bos_id=298), _Tok(300, pad_id=0, eos_id=1)
    pipe.text_encoder = CLIPTextEncoder(clip.state_dict(), num_layers=1, num_heads=1, eos_token_id=2)
    pipe.text_encoder_2 = T5Encoder(t5.state_dict(), num_layers=1, num_heads=2, d_kv=64)
    enc = PromptEncoder('flux', pipe, max_sequence_length=64)
    prompts = ['a fox', 'two red cubes on a table', 'sunset']
    names = write_cache(enc, prompts, str(tmp_path / 'cache'), latent_size=(16, 16, 16), batch=2, compress=False)
    assert names == ['00000000', '00000001', '00000002'] and (tmp_path / 'cache.jsonl').exists()
    ds = DATA.PromptEmbedCache(str(tmp_path / 'cache'), pad_seq_len=64)
    assert len(ds) == 3 and ds[1]['name'] == prompts[1]
    batch = DATA.collate([ds[0], ds[1]], device='cuda')
    online = enc.cond(prompts[:2], 8, 8)
    assert (batch['hp'], batch['wp']) == (8, 8)
    assert _rel(batch['prompt_embeds'], online['prompt_embeds'].cpu()) < 4e-3          # fp16 storage of bf16 values
    assert _rel(batch['pooled'], online['pooled'].cpu()) < 4e-3
