# ArcFlow-Qwen-Image distillation in the reference's config-file format with the hyper-parameters of configs/qwen/arcqwen_2nfe_k16.py +
# _ddp_train.py: 60 blocks, rank-256 adapters on img_mlp (all blocks) / txt_mlp (blocks 0..58) + the timestep embedder,
# true classifier-free guidance of the teacher at scale 4.0, decay 1000 -- and, on top (no key of the reference: it has no fp8 path), both
# networks' forwards on the fp8 MFMA with bf16 gradients = BASELINE.json configs[4].  tools/train.py also accepts the reference's own files.
name = 'qwen_distill_2nfe_fp8'

model = dict(
    diffusion=dict(
        type='ArcFlowImitationDataFree',
        policy_type='ArcFlow',
        denoising=dict(
            type='ArcQwenImageTransformer2DModel',
            num_gaussians=16, logweights_channels=4, in_channels=64, num_layers=60, attention_head_dim=128, num_attention_heads=24,
            joint_attention_dim=3584, use_lora=True, lora_rank=256, lora_dropout=0.05,
            lora_target_modules=['img_mlp.net.0.proj', 'img_mlp.net.2', 'timestep_embedder.linear_1', 'timestep_embedder.linear_2']
            + [f'transformer_blocks.{i}.txt_mlp.net.0.proj' for i in range(59)] + [f'transformer_blocks.{i}.txt_mlp.net.2' for i in range(59)]),
        flow_loss=dict(type='DiffusionMSELoss', rescale_cfg=dict(scale=30.0)),
        timestep_sampler=dict(type='ContinuousTimeStepSampler', shift=3.2)))

train_cfg = dict(num_decay_iters=1000, window_substeps=3, gm_dropout=0.1, num_intermediate_states=4, teacher_guidance_scale=4.0,
                 nfe=2, timestep_ratio=1.0, total_substeps=128, diffusion_grad_clip=50.0, diffusion_grad_clip_begin_iter=100,
                 teacher_fp8=True, student_fp8=True)
optimizer = {'diffusion': dict(type='AdamW8bit', lr=1e-4, betas=(0.9, 0.95), weight_decay=0.0,
                               paramwise_cfg=dict(custom_keys={'proj_out_loggamma': dict(lr_mult=0.1)}))}
lr_config = dict(policy='fixed', warmup='linear', warmup_iters=100, warmup_ratio=0.001)
runner = dict(ckpt_trainable_only=True, ckpt_fp16=True, ckpt_fp16_ema=True)
data = dict(train_dataloader=dict(samples_per_gpu=2))          # BASELINE.json configs[4]: global batch 16 on 8 GPUs
checkpoint_config = dict(interval=500, out_dir='checkpoints/')
total_iters = 15000
custom_hooks = [dict(type='ExponentialMovingAverageHookMod', start_iter=100, momentum_cfg=dict(gamma=7.0))]
resume_from = f'checkpoints/{name}/latest.pth'
