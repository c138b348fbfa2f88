#!/usr/bin/env python3
"""Headline benchmark: 1024x1024 images/s at 2 NFE on the ArcFlow-FLUX-12B architecture (BASELINE.json
configs[1]), one replica per GPU (images are independent units: weak scaling, no data-path collective).

A "step" = one image: 2 x { denoiser forward (19 double + 38 single MMDiT blocks, 4096 image + 512 text
tokens, bf16 MFMA) + analytic ArcFlow transport step } with the latents resident in HBM -- the loop of
lakonlab/pipelines/arcflux_pipeline.py:457-510.  Text encoders and the VAE are outside the hot path
 and outside the timed region; prompt embeddings are synthetic, weights are random-init of
the exact architecture (no network for checkpoints).

    python bench.py --gpus N --steps K --warmup W [--model flux|qwen] [--train]

N > 1: one replica per GPU.  Launched under torchrun (RANK / WORLD_SIZE in the environment) the script is one rank; launched
bare (`python bench.py --gpus N`) it re-executes itself under `python -m torch.distributed.run --nproc-per-node N` and
fails loudly when the box has fewer than N GPUs -- the line never reports n_gpus 1 for an N-GPU request.
--train times the distillation iteration instead (BASELINE.json configs[3] / [4]: FLUX 4 samples/GPU, Qwen 2 samples/GPU
with the true-CFG teacher; data parallel, the adapter-gradient all-reduce over RCCL is inside the timed region).

Prints ONE JSON line (rank 0) with the `roofline` object of the dominant kernel (the bf16 MFMA GEMM,
timed live with HIP events on the launch stream) and a `cpu_baseline` object (the fp32 CPU oracle's
FluxTransformerBlock on the host cores -- BASELINE.json configs[0] -- extrapolated by FLOPs).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TF = 2500.0        # dense, /opt/skills/guides/MI355X_MICROARCH.md:42
N_IMG, N_TXT, D_MODEL = 4096, 512, 3072
FLOPS_PER_FORWARD = 74.41e12      # 57 x (24 S D^2 + 4 S^2 D) + embed/head/mod, S = 4608


def build_flux_engine(model: str = 'flux', device: str = 'cuda', seed: int = 0):
    from arcflow_amd import MMDiTEngine
    from arcflow_amd.weights import random_packed
    if model == 'flux':
        eng = MMDiTEngine('flux', 19, 38, device=device)
        eng.bind_packed(random_packed('flux', 19, 38, device, seed=seed))
        T, joint = N_TXT, 4096
    else:
        eng = MMDiTEngine('qwen', 60, 0, joint_dim=3584, device=device)
        eng.bind_packed(random_packed('qwen', 60, 0, device, joint_dim=3584, seed=seed))
        T, joint = 128, 3584
    g = torch.Generator(device=device).manual_seed(42)
    x = torch.randn(1, N_IMG, 64, generator=g, device=device)
    ctx = (torch.randn(1, T, joint, generator=g, device=device) * 0.1).bfloat16()
    pooled = (torch.randn(1, 768, generator=g, device=device) * 0.1).bfloat16() if model == 'flux' else None
    guidance = torch.full((1,), 3.5, device=device) if model == 'flux' else None
    t = torch.ones(1, device=device)
    return eng, (x.bfloat16(), t, ctx, pooled, guidance, 64, 64)


def cpu_baseline(budget_s: float = 12.0):
    """BASELINE.json configs[0]: one FLUX double block, bs 1, 256 image + 77 text tokens, fp32, host cores."""
    from oracle import dit_ref as D
    cores = os.cpu_count() or 1
    cfg = D.FluxCfg(num_layers=1, num_single_layers=0)
    g = torch.Generator().manual_seed(0)
    w = {}
    p = 'transformer_blocks.0.'
    Dm = cfg.dim
    for nm, o, i in [('norm1.linear', 6 * Dm, Dm), ('norm1_context.linear', 6 * Dm, Dm)] + \
            [('attn.' + n, Dm, Dm) for n in ('to_q', 'to_k', 'to_v', 'add_q_proj', 'add_k_proj', 'add_v_proj', 'to_out.0', 'to_add_out')] + \
            [('ff.net.0.proj', 4 * Dm, Dm), ('ff.net.2', Dm, 4 * Dm), ('ff_context.net.0.proj', 4 * Dm, Dm), ('ff_context.net.2', Dm, 4 * Dm)]:
        w[p + nm + '.weight'] = torch.randn(o, i, generator=g) * 0.02
        w[p + nm + '.bias'] = torch.randn(o, generator=g) * 0.02
    for nm in ('norm_q', 'norm_k', 'norm_added_q', 'norm_added_k'):
        w[p + f'attn.{nm}.weight'] = 1 + 0.02 * torch.randn(128, generator=g)
    g1 = torch.Generator().manual_seed(1)
    img, txt = torch.randn(1, 256, Dm, generator=g1), torch.randn(1, 77, Dm, generator=g1)
    temb = torch.randn(1, Dm, generator=torch.Generator().manual_seed(2))
    cos, sin = D.flux_rope_tables(16, 16, 77)
    fn = lambda: D.flux_double_block(w, p, cfg, img, txt, temb, cos, sin)  # noqa: E731
    # the box may expose far more hardware threads than torch's CPU GEMM can use: calibrate the thread
    # count on one run each (a candidate slower than 4 s is abandoned), then sample the best one.
    best_thr, best_t = 1, float('inf')
    for thr in sorted({min(cores, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(thr)
        fn()
        t0 = time.perf_counter()
        fn()
        el = time.perf_counter() - t0
        if el < best_t:
            best_thr, best_t = thr, el
        if el > 4.0:
            break
    torch.set_num_threads(best_thr)
    cores_used = best_thr
    times = []
    t_end = time.time() + budget_s
    while (time.time() < t_end or len(times) < 3) and len(times) < 200:
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    S = 333
    flops = 24 * S * Dm * Dm + 4 * S * S * Dm + 2 * 2 * 6 * Dm * Dm      # 77.0 GFLOP
    gflops = flops / med / 1e9
    return {
        'value': gflops * 1e9 / (2 * FLOPS_PER_FORWARD), 'unit': 'images/s', 'cores': cores_used, 'host_cpus': cores, 'kind': 'port',
        'sample': f'oracle fp32 FluxTransformerBlock (configs[0]: 256 img + 77 txt tokens, bs 1), median of '
                  f'{len(times)} runs = {med*1e3:.1f} ms = {gflops:.0f} GFLOP/s, extrapolated by FLOPs to a '
                  f'148.8 TFLOP 2-NFE image',
        'block_ms': med * 1e3, 'gflops': gflops,
    }


# the bf16 block GEMMs of the forward: two instantiations of one kernel template -- 256x224 tiles for the N = 3072 / 12288 launches (96 per
# forward), 256x256 for the k|q|v^T and the single blocks' fused launches (57); the events cover every launch (AFX_GEMM_IMPL=2: gemm_kernel_v2<false>)
GEMM_KERNEL_NAME = 'afx::gemm_kernel_v3<8, 7, false, 0> + afx::gemm_kernel_v3<8, 8, false, 0>'      # (as rocprofv3 prints them: MI, NJ, CONV, PERSIST)
if os.environ.get('AFX_GEMM_IMPL', '3')[:1] == '2':
    GEMM_KERNEL_NAME = 'afx::gemm_kernel_v2<false>'
# --fp8: the one-wave-per-SIMD fp8 kernel, plain (row-scaled operands out of LayerNorm) and block-scaled (v_mfma_scale) instances; AFX_FP8_V3=0: the 8-phase kernel
FP8_KERNEL_NAME = 'afx::gemm_kernel_v2<true>' if os.environ.get('AFX_FP8_V3', '1')[:1] == '0' else 'afx::gemm_kernel_v3f8<8, 8, false> + afx::gemm_kernel_v3f8<8, 8, true> + afx::gemm_kernel_v3f8<7, 8, false> + afx::gemm_kernel_v3f8<7, 8, true>'
POWER_CAPPED_MFMA_TF = 1950.0


def _kernel_source_sha():
    """sha256 (first 16 hex digits) of the GEMM kernel's sources: profiles/traffic.json carries the value of the build its PMC pass measured."""
    import hashlib
    h = hashlib.sha256()
    for f in ('afx_gemm.hip', 'afx_common.h', 'afx_kernels.h'):
        with open(os.path.join(ROOT, 'arcflow_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _traffic(model):
    """HBM-side bytes per GEMM launch from the last committed PMC pass (profiles/traffic.json; tools/update_traffic.py writes it from the
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of tools/profile_round.sh).  bench.py cannot run rocprofv3 on itself, so the value is only
    printed when the pass measured THIS build of the kernel (the file carries the sha of the kernel sources); otherwise null."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            t = json.load(f)
        if t.get('kernel_source_sha16') != _kernel_source_sha():
            return None
        return t[model]['bytes_per_launch']
    except Exception:
        return None


def train_main(args, rank, world, dev, dist):
    """--train: one step = one distillation iteration (lakonlab/models/diffusions/arcflow.py:338-426 + the optimizer / EMA shell)
    on the per-GPU batch of the reference config; data parallel over the ranks (samples are independent: weak scaling), the
    single adapter-gradient all-reduce (RCCL) is inside the timed region and its exposed part is reported."""
    from arcflow_amd.train import ArcFlowDistiller, DistillConfig
    from arcflow_amd.weights import random_packed
    flux = args.model == 'flux'
    D = 3072
    nd, ns, joint, T = (19, 38, 4096, 512) if flux else (60, 0, 3584, 128)
    if args.blocks:                 # plumbing tests only (tests/test_multi_rank_plumbing.py): fewer blocks, the line's workload says so
        nd, ns = (int(v) for v in args.blocks.split(','))
    B = args.batch or (4 if flux else 2)
    packed = random_packed(args.model, nd, ns, dev, joint_dim=joint, seed=0)
    g = torch.Generator(device=dev).manual_seed(1)
    packed['teacher_head.weight'] = (torch.randn(64, D, generator=g, device=dev) * 0.02).bfloat16()
    packed['teacher_head.bias'] = torch.zeros(64, device=dev, dtype=torch.bfloat16)
    packed['norm_out.weight'] = packed['mod.weight'][-2 * D:].clone()
    packed['norm_out.bias'] = packed['mod.bias'][-2 * D:].clone()
    # configs/flux/arcflux_2nfe_k16.py / configs/qwen/arcqwen_2nfe_k16.py: rank-256 LoRA, lora_dropout 0.05; Qwen: true-CFG
    # teacher (scale 4.0, negative prompt), decay 1000
    kw = dict(lora_rank=256, lora_dropout=0.05, teacher_fp8=args.teacher_fp8, student_fp8=args.student_fp8)
    dc = DistillConfig(**kw) if flux else DistillConfig(teacher_guidance_scale=4.0, num_decay_iters=1000, **kw)
    eng = dict(num_double=nd, num_single=ns) if flux else dict(num_double=nd, joint_dim=joint)
    ds = ArcFlowDistiller(args.model, eng, None, dc, device=dev, packed=packed)
    cond = dict(prompt_embeds=(torch.randn(B, T, joint, device=dev, generator=g) * 0.1).bfloat16(), hp=64, wp=64)
    if flux:
        cond['pooled'] = (torch.randn(B, 768, device=dev, generator=g) * 0.1).bfloat16()
    else:
        cond['negative_prompt_embeds'] = (torch.randn(B, T, joint, device=dev, generator=g) * 0.1).bfloat16()
    rng = torch.Generator(device=dev).manual_seed(100 + rank)        # per-rank draws (train.py --diff_seed)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        info = ds.train_step(cond, B, rng=rng)
    barrier()
    t0 = time.perf_counter()
    exposed = 0.0
    for _ in range(args.steps):
        info = ds.train_step(cond, B, rng=rng)
        exposed += info['allreduce_exposed_ms']
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        from arcflow_amd.train import host_or_device
        tt = torch.tensor([dt], device=host_or_device(dist, dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    if rank == 0:
        # SURVEY 3.3 counts student 2 + teacher 8 (x2 with true CFG) + recompute 2 + backward ~4 = 16 (24) forward-equivalents per sample for the
        # REFERENCE, which checkpoints every block and recomputes its forward in the backward.  This engine keeps the forward's GEMM / attention
        # outputs in HBM instead, so the recompute is not executed and is not algorithmic work: `achieved` counts 14 (22).
        # ARCFLOW_TRAIN_RECOMPUTE=1 runs the reference's recompute schedule (A/B).
        recompute = os.environ.get('ARCFLOW_TRAIN_RECOMPUTE', '0') == '1'
        fwd_equiv_ref = 16 if flux else 24
        fwd_equiv = fwd_equiv_ref - 2
        per_fwd = FLOPS_PER_FORWARD if flux else 70.6e12
        sps = world * B * args.steps / dt
        ach = B * args.steps * fwd_equiv * per_fwd / dt / 1e12           # per GPU
        peak = MFMA_BF16_PEAK_TF
        # with fp8 forwards the forward-equivalents that run their GEMMs on the fp8 MFMA (5 PF dense) have a higher ceiling than 2.5 PF: blended peak =
        # work / (time at each part's own peak), GEMM share of a forward's flops 0.80 (FLUX) / 0.813 (Qwen-Image), attention and the backward at the bf16 peak
        f_gemm = 0.80 if flux else 0.813
        n_teacher = fwd_equiv - 6
        n_fp8 = (n_teacher if args.teacher_fp8 else 0) + (2 if args.student_fp8 else 0)
        t_units = n_fp8 * (f_gemm / (2 * peak) + (1 - f_gemm) / peak) + (fwd_equiv - n_fp8) / peak
        peak_blended = fwd_equiv / t_units
        line = {
            'metric': f'distillation samples/sec ({"ArcFlow-FLUX-12B" if flux else "ArcFlow-Qwen-Image-20B"} architecture, data-free '
                      f'trajectory matching, 2 student steps x 4 teacher states, LoRA r=256 + heads + norm_out trainable)',
            'value': sps, 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('fp8 e4m3 forward linears (teacher + student forward / recompute), bf16 gradients (configs[4])' if args.teacher_fp8 and args.student_fp8
                      else 'bf16 gradients and teacher, fp8 e4m3 student forward linears' if args.student_fp8
                      else 'bf16 student + gradients, fp8 e4m3 frozen-teacher linears (configs[4])' if args.teacher_fp8
                      else 'bf16 (fp32 master weights / gradients / AdamW moments)'),
            'data': 'synthetic (random-init weights of the exact architecture, synthetic prompt embeddings, data-free noise latents)',
            'config': {'workload': ('ArcFlow-FLUX distillation training (train_flux.sh), LoRA adapters' if flux else
                                    'ArcFlow-Qwen-20B distillation training (train_qwen.sh), true-CFG teacher')
                       + (f' -- REDUCED DEPTH {args.blocks} (plumbing test, not a measurement)' if args.blocks else ''),
                       'samples_per_gpu': B, 'global_batch': world * B, 'image_tokens': N_IMG, 'text_tokens': T,
                       'trainable_params': int(ds.params.numel()), 'lora_dropout': 0.05,
                       'parallelism': f'dp{world}: batch sharded over ranks, one flat-gradient all-reduce per iteration (RCCL), '
                                      f'sliced per block and overlapped with the last backward'},
            'roofline': {'bound': 'mfma', 'kernel': 'whole iteration (denoiser forward-equivalents, SURVEY 3.3)', 'achieved': ach,
                         'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
                         'peak_blended': peak_blended, 'frac_of_blended_peak': ach / peak_blended,
                         'peak_note': (f'{n_fp8} of the {fwd_equiv} forward-equivalents run their GEMMs on the fp8 MFMA (5 PF dense): `frac` is against the bf16 peak, '
                                       f'`frac_of_blended_peak` against work / (time of each part at its own peak)') if n_fp8 else 'all bf16: the blended peak is the bf16 peak',
                         'forward_equivalents_per_sample': fwd_equiv,
                         'note': f'algorithmic work = student 2 + teacher {fwd_equiv - 6} + backward 4 forward-equivalents; the reference additionally '
                                 f'recomputes 2 (its cost model: {fwd_equiv_ref}), which this engine ' + ('EXECUTES in this run (ARCFLOW_TRAIN_RECOMPUTE=1)' if recompute else 'replaces by keeping the forward outputs in HBM'),
                         'frac_by_reference_cost_model': ach / peak * fwd_equiv_ref / fwd_equiv},
            'backward_schedule': 'recompute every block from its checkpoint (the reference\'s)' if recompute else 'forward GEMM / attention outputs kept in HBM, element-wise work redone',

            'allreduce_exposed_ms_per_step': exposed / args.steps,
            'allreduce_bytes_per_step': int(ds.params.numel()) * 4 if world > 1 else 0,
            'max_mem_gb': torch.cuda.max_memory_allocated() / 2 ** 30, 'last_step': info,
        }
        try:      # the attention backward of ONE (sample, layer) at this workload's sequence length, alone on the chip (the kernel DESIGN 4.3 is about; HIP events, 10 calls)
            line['attn_backward'] = _attn_backward_probe(dev, T + N_IMG)
        except Exception as e:              # noqa: BLE001
            line['attn_backward'] = {'error': f'{type(e).__name__}: {e}'}
        return line
    return None


def _attn_backward_probe(dev, S, H=24, calls=10):
    from arcflow_amd import ops
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v, do = (torch.randn(1, S, H, 128, generator=g, device=dev).bfloat16() for _ in range(4))
    o, lse = ops.attention_fwd_lse(q, k, v)
    for _ in range(3):
        ops.attention_bwd(q, k, v, o, do, lse)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(calls):
        ops.attention_bwd(q, k, v, o, do, lse)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / calls
    tf = 5 * 2 * H * S * S * 128 / us * 1e-6
    return {'kernel': 'afx::b3::attn_bwd_fused3_kernel (+ attn_bwd_stats_kernel): generated dK / dV + dQ streams, one launch', 'tokens': S, 'heads': H,
            'us_per_call': us, 'achieved': tf, 'unit': 'TFLOP/s over the 5 algorithmic matmuls (7 executed)', 'peak': 2500.0, 'frac': tf / 2500.0, 'bound': 'mfma'}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default 5; 2 with --train)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed steps (default 2; 1 with --train)')
    ap.add_argument('--model', default='flux', choices=['flux', 'qwen'])
    ap.add_argument('--train', action='store_true', help='time the distillation iteration (configs[3] flux / configs[4] qwen) instead of inference')
    ap.add_argument('--blocks', default=None, help='--train: "double,single" block counts for plumbing tests (the line is then labelled REDUCED DEPTH)')
    ap.add_argument('--batch', type=int, default=None, help='--train: samples per GPU (default: 4 flux, 2 qwen, the reference configs)')
    ap.add_argument('--student-fp8', action='store_true', help='--train: the student forward / recompute linears on the fp8 MFMA, gradients bf16 (configs[4]); says so in dtype')
    ap.add_argument('--teacher-fp8', action='store_true', help='--train: frozen teacher forwards on the fp8 MFMA (configs[4]); the line says so in dtype')
    ap.add_argument('--streams', type=int, default=1, help='images in flight per GPU (one HIP stream + engine context each, shared weights); '
                                                           '1 = the canonical line, 2 fills the under-filled last rounds: +3.8 %% (r02)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='N = 1 FLUX run: skip the extra objects (Qwen inference, the two distillation iterations, prompt -> image)')
    ap.add_argument('--e2e', action='store_true', help='only the prompt -> image object of --model (encoders + 2 NFE + VAE)')
    ap.add_argument('--no-profile', action='store_true', help='do not record per-launch HIP events')
    ap.add_argument('--profile-stride', type=int, default=8, help='HIP event pair on one GEMM / attention launch in N inside the timed region (1: every launch, which costs 1.8 %% of the step)')
    ap.add_argument('--no-prepare-steps', action='store_true', help='recompute the AdaLN conditioning inside every transformer call (A/B)')
    ap.add_argument('--fp8', action='store_true', help='OPTIONAL reduced-precision mode: block linears on the fp8 MFMA (not the headline: the line says dtype fp8)')
    ap.add_argument('--master-port', type=int, default=None, help='rendezvous port of the self-launch (default: derived from the pid)')
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 2 if args.train else 5
    if args.warmup is None:
        args.warmup = 1 if args.train else 2
    return args


def launcher_cmd(args, argv, environ=None, device_count=None):
    """The command `python bench.py --gpus N` re-executes itself with when it is NOT already a torchrun rank, or None when
    this process should run the bench itself.  Raises when the request cannot be honoured (never a silent n_gpus=1)."""
    environ = os.environ if environ is None else environ
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if 'WORLD_SIZE' in environ:                      # already one rank of a torchrun launch
        world = int(environ['WORLD_SIZE'])
        if world != args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}')
        return None
    if args.gpus == 1:
        return None
    if device_count is None:
        device_count = torch.cuda.device_count()
    if device_count < args.gpus and environ.get('ARCFLOW_DIST_ONE_DEVICE', '0') != '1':
        raise SystemExit(f'bench.py: --gpus {args.gpus} requested but this box exposes {device_count} GPU(s)')
    port = args.master_port or (29500 + os.getpid() % 2000)
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), *argv]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    cmd = launcher_cmd(args, argv)
    if cmd is not None:
        import subprocess
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        raise SystemExit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    dist = None
    if world > 1:
        from arcflow_amd.train import init_distributed
        dist, dev = init_distributed(local_rank)       # RCCL, rank r on GPU r (one-GPU test boxes: see its docstring)
    else:
        torch.cuda.set_device(0)
        dev = 'cuda:0'
    if args.e2e:
        line = {'e2e': {args.model: e2e_main(args.model, dev, steps=args.steps, warmup=args.warmup)}} if rank == 0 else None
    elif args.train:
        line = train_main(args, rank, world, dev, dist)
    else:
        line = infer_main(args, args.model, rank, world, dev, dist)
        if rank == 0 and world == 1 and args.model == 'flux' and not args.no_extras and not args.fp8 and args.streams == 1:
            # The driver times ONE default run: after the FLUX headline (whose timed region is over) the same process measures the
            # other configs BASELINE.json names -- Qwen-Image inference (configs[2]), one FLUX distillation iteration (configs[3]) and one Qwen-Image
            # distillation iteration with the fp8 forwards (configs[4]) --
            # and attaches them as extra objects.  The headline's value / ms_per_step / steps are untouched.
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            ex = argparse.Namespace(**vars(args))
            KEYS_T = ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'config', 'roofline', 'attn_backward', 'backward_schedule',
                      'allreduce_exposed_ms_per_step', 'allreduce_bytes_per_step', 'max_mem_gb')

            def extra(name, fn):
                """An extra object must never cost the headline its line: a failure is recorded under its key and the run goes on."""
                try:
                    line[name] = fn()
                except Exception as e:              # noqa: BLE001
                    import traceback
                    line[name] = {'error': f'{type(e).__name__}: {e}', 'where': traceback.format_exc().strip().splitlines()[-3:]}
                gc.collect()
                torch.cuda.empty_cache()

            def qwen_infer():
                ex.steps, ex.warmup = min(args.steps, 5), min(args.warmup, 2)
                q = infer_main(ex, 'qwen', rank, world, dev, dist)
                return {k: q[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'config', 'mfma_frac_end_to_end', 'roofline',
                                          'roofline_attention') if k in q}

            def train(model, fp8):
                # configs[4]: Qwen-Image distillation, true-CFG teacher, "fp8 MFMA fwd + bf16 grads" (teacher and student forwards on the e4m3
                # MFMA; says so in its dtype -- reduced precision, an extra object, never the headline)
                ex.steps, ex.warmup, ex.model, ex.batch, ex.teacher_fp8, ex.student_fp8 = 2, 1, model, None, fp8, fp8
                t = train_main(ex, rank, world, dev, dist)
                return {k: t[k] for k in KEYS_T if k in t}
            extra('qwen', qwen_infer)
            extra('train_flux', lambda: train('flux', False))
            extra('train_qwen_fp8', lambda: train('qwen', True))
            # prompt -> image (encoders + 2 NFE + VAE), both families: BASELINE.md section 2 "reported separately"
            line['e2e'] = {}
            for m in ('flux', 'qwen'):
                try:
                    line['e2e'][m] = e2e_main(m, dev)
                except Exception as e:              # noqa: BLE001
                    line['e2e'][m] = {'error': f'{type(e).__name__}: {e}'}
                gc.collect()
                torch.cuda.empty_cache()
        if rank == 0 and not args.no_cpu_baseline and world == 1:
            try:
                line['cpu_baseline'] = cpu_baseline()
            except Exception as e:                  # noqa: BLE001  (the headline line must still be printed)
                line['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def e2e_main(model, dev, steps=5, warmup=2):
    """Prompt -> image at 1024^2 on one GPU (BASELINE.md section 2 promised it next to the denoiser-only headline): the prompt
    encoders (FLUX: T5-XXL at 512 tokens + CLIP-L at 77; Qwen-Image: the Qwen2.5-VL-7B language model at 34 template + 128 prompt
    tokens), the 2-NFE denoiser loop and the VAE decode, all on the HIP library, random-init weights of the released sizes.
    Starts from token ids (tokenisation is host-side string work on the snapshot's vocabulary files).  Three numbers:
    `from_prompt_embeds` (pipe(prompt_embeds=...): denoiser + VAE), `from_token_ids` (pipe(prompt=...) minus the tokenizer),
    `from_token_ids_vae_overlapped` (throughput mode: the VAE decode of image i on a second stream under the encoders + denoiser
    of image i + 1).  Reference sequence: lakonlab/pipelines/arcflux_pipeline.py:385-534, arcqwen_pipeline.py:346-481."""
    from arcflow_amd import ops, synthetic
    from arcflow_amd.text_encoders import CLIPTextEncoder, Qwen25TextEncoder, T5Encoder
    from arcflow_amd.vae import AutoencoderKLDecoder, AutoencoderKLQwenImageDecoder
    eng, (_, _, ctx0, pooled0, guidance, hp, wp) = build_flux_engine(model, dev)
    sig = [1.0, 0.7619047619, 0.0]
    tv = [torch.full((1,), s, device=dev) for s in sig[:2]]
    lat = torch.randn(1, N_IMG, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(42))
    if model == 'flux':
        t5, clip = T5Encoder(synthetic.t5_state_dict(dev)), CLIPTextEncoder(synthetic.clip_state_dict(dev), eos_token_id=2)
        vae = AutoencoderKLDecoder(synthetic.vae_kl_decoder_state_dict(dev), (128, 256, 512, 512))
        ids5, idsc = torch.randint(0, 32000, (1, 512)), torch.randint(0, 49000, (1, 77))

        def encode():
            return t5(ids5), clip(idsc)[1]
        enc_desc = 'T5-XXL encoder 512 tokens + CLIP-L text model 77 tokens'
    else:
        qw = Qwen25TextEncoder(synthetic.qwen25_state_dict(dev))
        vae = AutoencoderKLQwenImageDecoder(synthetic.vae_qwen_decoder_state_dict(dev), [0.0] * 16, [1.0] * 16)
        idsq = torch.randint(0, 150000, (1, 34 + 128))

        def encode():
            return qw(idsq)[:, 34:].contiguous(), None        # the 34 template tokens are dropped (arcqwen_pipeline.py prompt template)
        enc_desc = 'Qwen2.5-VL-7B language model, 34 template + 128 prompt tokens'

    def denoise(pe, pooled):
        x = lat
        prep = eng.prepare_steps(sig[:2], pooled, guidance, 1, N_IMG, int(pe.shape[1]))
        for i in range(2):
            out = eng(x.bfloat16(), tv[i], pe, pooled, guidance, hp, wp, prepared_step=i if prep else None)
            x = ops.arcflow_step(x, out.means, out.logweights, out.loggammas, sig[i], sig[i], sig[i + 1])
        return x

    def timed(fn, n=steps, w=warmup):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, r

    pe0, pl0 = encode()
    enc_ms, _ = timed(encode)
    den_ms, x = timed(lambda: denoise(pe0, pl0))
    vae_ms, img = timed(lambda: vae.decode_packed(x, hp, wp))
    assert torch.isfinite(img.float()).all() and img.shape[-2:] == (1024, 1024), img.shape
    emb_ms, _ = timed(lambda: vae.decode_packed(denoise(pe0, pl0), hp, wp))
    ids_ms, _ = timed(lambda: vae.decode_packed(denoise(*encode()), hp, wp))
    # throughput mode: image i's decode on a second stream under image i + 1's encoders + denoiser
    side = torch.cuda.Stream(device=dev)
    main_s = torch.cuda.current_stream()

    def pipelined():
        xx = denoise(*encode())
        ev = torch.cuda.Event()
        ev.record(main_s)
        xx.record_stream(side)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            out = vae.decode_packed(xx, hp, wp)
        return out
    ovl_ms, _ = timed(pipelined, n=max(steps, 6))
    main_s.wait_stream(side)

    def obj(ms):
        return {'value': 1e3 / ms, 'unit': 'images/s', 'ms_per_image': ms}
    return {'workload': f'prompt -> 1024x1024 image, 2 NFE, bs 1, one GPU, one image at a time ({enc_desc}; '
                        f'{"FLUX-12B" if model == "flux" else "Qwen-Image-20B"} denoiser; '
                        f'{"AutoencoderKL" if model == "flux" else "AutoencoderKLQwenImage"} decoder), random-init weights of the released sizes',
            'from_prompt_embeds': obj(emb_ms), 'from_token_ids': obj(ids_ms), 'from_token_ids_vae_overlapped': obj(ovl_ms),
            'stages_ms': {'text_encoders': enc_ms, 'denoiser_2nfe': den_ms, 'vae_decode': vae_ms}, 'steps': steps, 'warmup': warmup}


def infer_main(args, model, rank, world, dev, dist):
    """One inference measurement (W warm-up + K timed images, barrier + synchronize on both sides, MAX over ranks).  Returns the line
    (rank 0) or None."""
    from arcflow_amd import ops
    from arcflow_amd.schedule import FlowMatchEulerDiscreteScheduler, retrieve_raw_timesteps

    eng, (x0, t, ctx, pooled, guidance, hp, wp) = build_flux_engine(model, dev, seed=rank)
    if args.fp8:
        eng.enable_fp8()
    engines, streams = [eng], [torch.cuda.current_stream()]
    for _ in range(1, max(1, args.streams)):          # more images in flight: one context (workspace) per stream, the SAME weight tensors
        from arcflow_amd import MMDiTEngine
        e2 = MMDiTEngine(eng.family, eng.num_double, eng.num_single, joint_dim=eng.joint_dim, device=dev)
        e2.bind_packed(eng._weights)
        if args.fp8:
            e2.enable_fp8()
        engines.append(e2)
        streams.append(torch.cuda.Stream(device=dev))
    raw, counts, _ = retrieve_raw_timesteps(2, 128, 1.0)
    sch = FlowMatchEulerDiscreteScheduler(shift=3.2)
    ts = sch.set_timesteps(sigmas=raw)
    sig = [float(ts[0]) / 1000, float(ts[counts[0]]) / 1000, 0.0]
    tvec = [torch.full((1,), s, device=dev) for s in sig[:2]]
    lat = torch.randn(1, N_IMG, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(42))

    n_img = [0]

    def one_image():
        k = n_img[0] % len(engines)
        n_img[0] += 1
        with torch.cuda.stream(streams[k]):
            x = lat
            # as the pipeline's loop does (arcflow_amd/pipelines/arcflux_pipeline.py::_denoise): the conditioning of both steps in
            # one pass over the modulation matrix -- per image, inside the timed region
            prep = engines[k].prepare_steps(sig[:2], pooled, guidance, 1, N_IMG, int(ctx.shape[1])) if not args.no_prepare_steps else False
            for i in range(2):
                out = engines[k](x.bfloat16(), tvec[i], ctx, pooled, guidance, hp, wp, prepared_step=i if prep else None)
                x = ops.arcflow_step(x, out.means, out.logweights, out.loggammas, sig[i], sig[i], sig[i + 1])
        return x

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_image()
    prof = not args.no_profile and len(engines) == 1      # per-launch durations overlap with several images in flight
    stride = max(1, getattr(args, 'profile_stride', 8))
    eng.profile(stride if prof else False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_image()
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(res).all()
    if dist is not None:
        from arcflow_amd.train import host_or_device
        tt = torch.tensor([dt], device=host_or_device(dist, dev))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    # the analytic transport step alone (SURVEY 8d asks for its GB/s): 50 launches between two events on the launch stream -- includes
    # the ~1-2 us boundary between dependent launches, so it under-states the kernel (rocprof: profiles/r03_*)
    step_gbs = None
    if rank == 0:
        outs = engines[0](lat.bfloat16(), tvec[0], ctx, pooled, guidance, hp, wp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            ops.arcflow_step(lat, outs.means, outs.logweights, outs.loggammas, sig[0], sig[0], sig[1])
        e0.record()
        for _ in range(50):
            ops.arcflow_step(lat, outs.means, outs.logweights, outs.loggammas, sig[0], sig[0], sig[1])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        step_bytes = N_IMG * (16 * 64 + 16 * 4 + 15 * 4) * 2 + 2 * N_IMG * 64 * 4       # bf16 mixture in, fp32 latents in + out: 11.5 MB
        step_gbs = {'bound': 'hbm', 'kernel': 'afx::arcflow_step_k16_kernel', 'achieved': step_bytes / us * 1e-3, 'peak': 8000.0, 'unit': 'GB/s',
                    'frac': step_bytes / us * 1e-3 / 8000.0, 'algorithmic_bytes_per_launch': step_bytes, 'avg_launch_us': us,
                    'note': '50 back-to-back launches between two events (includes the inter-launch boundary)'}
    gemm_ms, gemm_n, gemm_fl = eng.profile_read(0) if prof else (0.0, 0, 0.0)
    att_ms, att_n, att_fl = eng.profile_read(1) if prof else (0.0, 0, 0.0)
    eng.profile(False)

    if rank == 0:
        flops_img = 2 * (FLOPS_PER_FORWARD if model == 'flux' else 70.6e12)
        ips = world * args.steps / dt
        line = {
            'metric': f'1024x1024 images/sec @ 2 NFE ({"FLUX-12B" if model == "flux" else "Qwen-Image-20B"} '
                      f'architecture, denoiser + ArcFlow integrator)',
            'value': ips, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('fp8 e4m3 block linears (weights: one scale per output channel; activations: one scale per row out of LayerNorm-modulate, E8M0 block scales '
                      'per row x 128 columns out of the GEMM / attention epilogues) + bf16 attention / embedders / head: REDUCED PRECISION, not the headline') if args.fp8 else 'bf16', 'data': 'synthetic (random-init weights of the exact architecture, synthetic prompt '
                                     'embeddings, seeded noise latents)',
            'config': {'workload': 'ArcFlow-FLUX-12B 2-NFE inference, 1024x1024, bs=1 per GPU' if model == 'flux'
                       else 'ArcFlow-Qwen-Image-20B 2-NFE inference, 1024x1024, bs=1 per GPU, T=128',
                       'image_tokens': N_IMG, 'text_tokens': int(ctx.shape[1]), 'nfe': 2, 'sigmas': sig,
                       'parallelism': f'{world} independent replica(s), no collective' + (f', {len(engines)} images in flight per GPU (HIP streams)' if len(engines) > 1 else ''),
                       'lora': 'merged into base weights'},
            'mfma_frac_end_to_end': flops_img * ips / world / (MFMA_BF16_PEAK_TF * 1e12),
        }
        if prof and gemm_n:
            ach = gemm_fl / (gemm_ms * 1e-3) / 1e12
            line['roofline'] = {
                'bound': 'mfma', 'kernel': FP8_KERNEL_NAME if args.fp8 else GEMM_KERNEL_NAME, 'achieved': ach, 'peak': MFMA_BF16_PEAK_TF * (2 if args.fp8 else 1),
                'unit': 'TFLOP/s', 'frac': ach / (MFMA_BF16_PEAK_TF * (2 if args.fp8 else 1)), 'traffic': None if args.fp8 else _traffic(model),
                'launches': gemm_n, 'avg_launch_us': gemm_ms * 1e3 / gemm_n,
                'algorithmic_flops_per_launch': gemm_fl / gemm_n,
                'sampling': f'HIP event pair on 1 launch in {stride} of the timed region (every launch position is visited: 211 launches per forward)',
                'share_of_step_time': gemm_ms * stride * 1e-3 / dt,
                # what the power cap leaves of `peak`: a loop of nothing but 16x16x32 bf16 MFMAs on random operands runs at
                # 1.93-1.98 GHz / 1927-1984 TFLOP/s on this part (tools/gemm_trace.hip mfma_burn_rand, profiles/r02s_gemm_trace.txt)
                'power_capped_mfma_peak': POWER_CAPPED_MFMA_TF,
                'frac_of_power_capped_peak': ach / (POWER_CAPPED_MFMA_TF * (2 if args.fp8 else 1)),
            }
            if att_n:
                a2 = att_fl / (att_ms * 1e-3) / 1e12
                line['roofline_attention'] = {
                    'bound': 'mfma', 'kernel': 'afx::a3::attention_v3_kernel', 'achieved': a2, 'peak': MFMA_BF16_PEAK_TF,
                    'unit': 'TFLOP/s', 'frac': a2 / MFMA_BF16_PEAK_TF, 'launches': att_n,
                    'avg_launch_us': att_ms * 1e3 / att_n, 'share_of_step_time': att_ms * stride * 1e-3 / dt}
        line['roofline_step'] = step_gbs
        return line
    return None


if __name__ == '__main__':
    main()
