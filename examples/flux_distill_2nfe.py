# Minimal training config in the reference's config-file format (SURVEY.md appendix C lists the hyper-parameters of
# configs/flux/arcflux_2nfe_k16.py + _ddp_train.py); tools/train.py also accepts the reference's own files unchanged.
name = 'flux_distill_2nfe'

model = dict(
    diffusion=dict(
        type='ArcFlowImitationDataFree',
        policy_type='ArcFlow',
        denoising=dict(
            type='ArcFluxTransformer2DModel',
            num_gaussians=16, logweights_channels=4, in_channels=64, num_layers=19, num_single_layers=38,
            attention_head_dim=128, num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
            guidance_embeds=True, use_lora=True, lora_rank=256, lora_dropout=0.05,
            lora_target_modules=['proj_mlp', 'proj_out', 'ff.net.0.proj', 'ff.net.2', 'ff_context.net.0.proj', 'ff_context.net.2',
                                 'timestep_embedder.linear_1', 'timestep_embedder.linear_2']),
        flow_loss=dict(type='DiffusionMSELoss', rescale_cfg=dict(scale=30.0)),
        timestep_sampler=dict(type='ContinuousTimeStepSampler', shift=3.2)))

train_cfg = dict(num_decay_iters=2000, window_substeps=3, gm_dropout=0.1, num_intermediate_states=4,
                 distilled_guidance_scale=3.5, nfe=2, timestep_ratio=1.0, total_substeps=128,
                 diffusion_grad_clip=50.0, diffusion_grad_clip_begin_iter=100)
optimizer = {'diffusion': dict(type='AdamW8bit', lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0,
                               paramwise_cfg=dict(custom_keys={'proj_out_loggamma': dict(lr_mult=0.1)}))}
lr_config = dict(policy='fixed', warmup='linear', warmup_iters=100, warmup_ratio=0.001)
runner = dict(ckpt_trainable_only=True, ckpt_fp16=True, ckpt_fp16_ema=True)
data = dict(train_dataloader=dict(samples_per_gpu=4))
checkpoint_config = dict(interval=500, out_dir='checkpoints/')
total_iters = 10000
custom_hooks = [dict(type='ExponentialMovingAverageHookMod', start_iter=100, momentum_cfg=dict(gamma=7.0))]
resume_from = f'checkpoints/{name}/latest.pth'
