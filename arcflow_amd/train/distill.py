"""Data-free trajectory-matching distillation step (the reference's ``ArcFlowImitationDataFree``,
lakonlab/models/diffusions/arcflow.py:338-426, driven by ``train_fwd_bwd`` lakonlab/models/base_diffusion.py:14-62
and ``BaseModel.train_step`` lakonlab/models/base.py:76-103,162-189) on one MI355X per process.

Per iteration and student step (nfe = 2):
    student forward (HIP engine)  ->  K-component momentum policy
    piid_segment_momentum (arcflow.py:120-209): 4 x { roll the detached (GM-dropout) policy to t_a,
        teacher forward at (x_a, t_a), predicted mean velocity of the FULL policy over [t_b - window, t_a],
        teacher Euler roll to t_b }, MSE x 30 x 0.5 x segment size, roll to the segment end
    backward of the policy math -> head logits -> {velocity heads, norm_out.linear}  (fp32 gradients)
then: gradient all-reduce (RCCL, one exchange per iteration, launched slice by slice while the last backward runs), global-norm clip (50 from
iteration 100, non-finite -> skip), AdamW (loggamma rows lr x 0.1, linear warm-up), Karras EMA of the trainables.

Every tensor op on [B,N,*] data is a HIP kernel from libarcflow_hip (arcflow_amd.ops); torch supplies device memory,
the RNG draws and the tiny [B]-sized schedule arithmetic.

Trainable set = the reference's ``freeze_exclude`` (configs/flux/arcflux_2nfe_k16.py:20-25): the three velocity heads,
norm_out and (``lora_rank`` > 0) the rank-r LoRA adapters incl. the timestep-embedder pair, all in ONE flat fp32 buffer;
both student steps accumulate into ONE gradient buffer whose slices are all-reduced once each (train/reducer.py).
Batches above 4 samples per GPU run as micro-batches of <= 4 (the engine's grouped launches hold 4 samples; the
reference micro-batches the same way, ``grad_accum_batch_size``, configs/flux/_ddp_train.py:14).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import os

import torch

from .. import ops
from ..engine import MMDiTEngine
from ..weights import pack_flux, pack_head, pack_qwen
from .reducer import GradReducer
from .trunk import LoraTrunk


@dataclass
class DistillConfig:
    """train_cfg / optimizer values of configs/flux/arcflux_2nfe_k16.py:89-99 and _ddp_train.py:13-31."""
    nfe: int = 2
    timestep_ratio: float = 1.0
    total_substeps: int = 128
    window_substeps: int = 3
    gm_dropout: float = 0.1
    num_intermediate_states: int = 4
    num_decay_iters: int = 2000
    shift: float = 3.2
    eps: float = 1e-4
    loss_scale: float = 30.0
    guidance: float = 3.5                 # distilled guidance fed to the student (FLUX; distilled_guidance_scale)
    teacher_guidance: Optional[float] = None   # teacher_distilled_guidance_scale (arcflux_2nfe_k16.py:94); None = same as the student's
    teacher_guidance_scale: float = 1.0   # > 1: true CFG with negative prompt embeds (Qwen: 4.0)
    lr: float = 1e-4
    betas: tuple = (0.9, 0.95)
    weight_decay: float = 0.0
    optimizer: str = 'adamw'          # 'adamw': fp32 moments (the pinned math); 'adamw8bit': block-wise 8-bit moments as the
                                      # reference's bitsandbytes AdamW8bit keeps them (_ddp_train.py:18-26), groups < 4096 values stay fp32
    loggamma_lr_mult: float = 0.1
    warmup_iters: int = 100
    warmup_ratio: float = 0.001
    grad_clip: float = 50.0
    grad_clip_begin_iter: int = 100
    ema_gamma: float = 7.0
    ema_start_iter: int = 100
    lora_rank: int = 0                    # 0: train heads + norm_out only; 256 in the reference configs
    lora_dropout: float = 0.0             # peft lora_dropout on the adapters' input (0.05 in the reference configs)
    teacher_fp8: bool = False             # BASELINE.json configs[4]: frozen teacher forwards on the fp8 MFMA (student + grads stay bf16)
    student_fp8: bool = False             # configs[4] "fp8 MFMA fwd + bf16 grads": the student's block linears run their forward (and the
                                          # backward's recompute) on the fp8 MFMA, dgrad / LoRA gradients stay bf16 (train/trunk.py enable_fp8)


def warp(t: torch.Tensor, shift: float) -> torch.Tensor:
    """sigma = s t / (1 + (s-1) t)   (ContinuousTimeStepSampler.warp_t, sampler.py:46-48)."""
    return shift * t / (1 + (shift - 1) * t)


class ArcFlowDistiller:
    def __init__(self, family: str, engine_kwargs: dict, state_dict: Optional[Dict[str, torch.Tensor]], cfg: DistillConfig,
                 device='cuda', process_group=None, packed: Optional[Dict[str, torch.Tensor]] = None, init_seed: int = 1234):
        """state_dict: diffusers keys of the teacher (with ``proj_out``) plus the student heads ``proj_out_*``.
        Alternatively ``packed``: an already fused weight set on the device (arcflow_amd.weights.random_packed) that
        also carries ``teacher_head.{weight,bias}`` and ``norm_out.{weight,bias}`` -- synthetic-weight benchmarks."""
        self.cfg, self.family, self.device = cfg, family, torch.device(device)
        kw = dict(engine_kwargs)
        nd, ns = kw.pop('num_double'), kw.pop('num_single', 0)
        self.student = MMDiTEngine(family, nd, ns, device=device, **kw)
        self.teacher = MMDiTEngine(family, nd, ns, device=device, teacher_head=True, **kw)
        K, C, L = self.student.num_gaussians, self.student.in_channels, self.student.logweights_channels
        if packed is not None:
            packed = dict(packed)
            th_w, th_b = packed.pop('teacher_head.weight'), packed.pop('teacher_head.bias')
            no_w32, no_b32 = packed.pop('norm_out.weight').float(), packed.pop('norm_out.bias').float()
        else:
            if family == 'flux':
                packed = pack_flux(state_dict, nd, ns, self.device, K, C, L, self.student.guidance_embeds)
            else:
                packed = pack_qwen(state_dict, nd, self.device, K, C, L)
            th_w, th_b = pack_head(state_dict, self.device, K, C, L, teacher=True)
            no_w32 = state_dict['norm_out.linear.weight'].to(self.device, torch.float32)
            no_b32 = state_dict['norm_out.linear.bias'].to(self.device, torch.float32)
        # frozen trunk weights exist ONCE and are bound into both contexts (tie_untrained_submodules, utils/misc.py:116-133)
        t_packed = dict(packed)
        t_packed['head.weight'], t_packed['head.bias'] = th_w, th_b
        self.teacher.bind_packed(t_packed)
        if cfg.teacher_fp8:
            self.teacher.enable_fp8()
        D = self.student.dim
        self.D, self.K, self.C, self.L = D, K, C, L
        self.head_n = packed['head.weight'].shape[0]
        # ---- trainable set, flat fp32: [head.weight | head.bias | norm_out.weight | norm_out.bias] --------------
        no_w, no_b = no_w32, no_b32
        sizes = [self.head_n * D, self.head_n, 2 * D * D, 2 * D]
        self._off = [0]
        for s in sizes:
            self._off.append(self._off[-1] + s)
        n_lora = LoraTrunk.num_params(family, nd, ns, D, cfg.lora_rank) if cfg.lora_rank > 0 else 0
        n = self._off[-1] + n_lora
        self.params = torch.empty(n, dtype=torch.float32, device=self.device)
        self._view(self.params, 0).copy_(packed['head.weight'].float().flatten())
        self._view(self.params, 1).copy_(packed['head.bias'].float())
        self._view(self.params, 2).copy_(no_w.flatten())
        self._view(self.params, 3).copy_(no_b)
        if cfg.optimizer not in ('adamw', 'adamw8bit'):
            raise ValueError(f'optimizer must be adamw or adamw8bit, got {cfg.optimizer!r}')
        self.opt8 = {}                                 # (a, b) -> ops.AdamW8bitState of that learning-rate group (adamw8bit)
        if cfg.optimizer == 'adamw8bit':               # fp32 moments only for the groups bitsandbytes keeps in 32 bit (< 4096 values)
            self.exp_avg = self.exp_avg_sq = None
            self._small = {}
        else:
            self.exp_avg = torch.zeros_like(self.params)
            self.exp_avg_sq = torch.zeros_like(self.params)
        self.ema = self.params.clone()
        self.grad = torch.zeros_like(self.params)      # ONE buffer: both student steps accumulate into it
        self.grads = [self.grad]                       # (kept for callers that index the summed gradient as grads[0])
        # bf16 working copies the engine reads
        self.w_head = packed['head.weight']
        self.b_head = packed['head.bias']
        self.w_no = torch.empty(2 * D, D, dtype=torch.bfloat16, device=self.device)
        self.b_no = torch.empty(2 * D, dtype=torch.bfloat16, device=self.device)
        packed['mod_final.weight'], packed['mod_final.bias'] = self.w_no, self.b_no
        self._sync_working_copies()
        self.trunk = None
        if cfg.lora_rank > 0:      # the adapters run UN-merged through the trunk on the frozen tensors (shared with the teacher) + their own B columns;
                                   # the student ENGINE stays bound to the frozen weights until student_forward() asks the trunk for W + B A copies
            self.trunk = LoraTrunk(self.student, packed, cfg.lora_rank, self.params, self._off[4],
                                   generator=torch.Generator(device=self.device).manual_seed(init_seed))
            self.ema[self._off[4]:].copy_(self.params[self._off[4]:])
        self.student.bind_packed(packed)
        if cfg.student_fp8:
            if self.trunk is not None:
                self.trunk.enable_fp8(shared=self.teacher._weights if cfg.teacher_fp8 else None)
            else:
                self.student.enable_fp8()
        self._ckpt = None
        self.reducer = GradReducer(process_group)
        self.sync_module_states()
        self.iteration = 0
        self.opt_steps = 0
        self._norm_acc = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._loss_acc = torch.zeros(1, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ helpers
    def sync_module_states(self, src: int = 0) -> None:
        """Data parallel: every rank takes rank `src`'s trainables and EMA (the flat fp32 buffers), as torch DDP broadcasts the
        wrapped module's state at construction (lakonlab/parallel/ddp_wrapper.py:19-25).  Then the bf16 working copies / merged LoRA
        weights are rebuilt and the ranks' buffers are checksummed against each other.  No-op on one rank.
        The optimizer moments, `opt_steps` and `iteration` are NOT broadcast (DDP does not either): a checkpoint has to be loaded on
        EVERY rank, which is what tools/train.py does -- loading on one rank and calling this would leave the learning-rate schedule,
        the bias correction and the moments rank-local."""
        red = self.reducer
        if red._skip_single:
            return
        red.broadcast_(self.params, src)
        red.broadcast_(self.ema, src)
        self._sync_working_copies()
        if self.trunk is not None:
            self.trunk.refresh()
        red.check_consistent(self.params)

    def _view(self, flat: torch.Tensor, i: int) -> torch.Tensor:
        return flat[self._off[i]:self._off[i + 1]]

    def optimizer_groups(self):
        """[(a, b, is_loggamma)]: ranges of the flat buffer the optimizer treats as units.  fp32 AdamW is element-wise, so only the
        learning-rate split matters there (5 ranges).  adamw8bit follows bitsandbytes' granularity -- one state PER PARAMETER TENSOR
        (optimizer/builder.py:11-24): its 256-value absmax blocks start at tensor starts and never straddle two tensors, and a tensor
        below 4096 values (min_8bit_size: every bias of the heads / norm_out) keeps fp32 moments.  Cached: the layout is fixed."""
        if getattr(self, '_opt_groups', None) is not None:
            return self._opt_groups
        K, C, L, D = self.K, self.C, self.L, self.D
        n1, n2 = K * C, K * C + K * L                              # head rows: means | logweights | loggamma
        hw, hb = self._off[0], self._off[1]
        if self.cfg.optimizer != 'adamw8bit':
            g = [(hw, hw + n2 * D, False), (hw + n2 * D, self._off[1], True), (hb, hb + n2, False), (hb + n2, self._off[2], True),
                 (self._off[2], self.params.numel(), False)]
        else:
            g = [(hw, hw + n1 * D, False), (hw + n1 * D, hw + n2 * D, False), (hw + n2 * D, self._off[1], True),
                 (hb, hb + n1, False), (hb + n1, hb + n2, False), (hb + n2, self._off[2], True),
                 (self._off[2], self._off[3], False), (self._off[3], self._off[4], False)]
            if self.trunk is not None:
                r = self.trunk.r
                for sp in self.trunk.specs:
                    g += [(sp.off_a, sp.off_a + r * sp.in_f, False), (sp.off_b, sp.off_b + sp.out_f * r, False)]
            g.sort()
            assert g[0][0] == 0 and g[-1][1] == self.params.numel() and all(x[1] == y[0] for x, y in zip(g, g[1:])), \
                'optimizer groups must tile the flat buffer'
        self._opt_groups = [x for x in g if x[1] > x[0]]
        return self._opt_groups

    def _sync_working_copies(self):
        ops.cast_bf16(self._view(self.params, 0), self.w_head.view(-1))
        ops.cast_bf16(self._view(self.params, 1), self.b_head)
        ops.cast_bf16(self._view(self.params, 2), self.w_no.view(-1))
        ops.cast_bf16(self._view(self.params, 3), self.b_no)

    def trainable_state_dict(self, ema: bool = False) -> Dict[str, torch.Tensor]:
        """diffusers-keyed trainables (the adapter file layout of export_arcflow_to_diffusers.py:100-127)."""
        src = self.ema if ema else self.params
        K, C, L, D = self.K, self.C, self.L, self.D
        hw = self._view(src, 0).view(self.head_n, D)
        hb = self._view(src, 1)
        n1, n2 = K * C, K * C + K * L
        n3 = n2 + (K - 1) * L
        return {
            'proj_out_means.weight': hw[:n1].clone(), 'proj_out_means.bias': hb[:n1].clone(),
            'proj_out_logweights.weight': hw[n1:n2].clone(), 'proj_out_logweights.bias': hb[n1:n2].clone(),
            'proj_out_loggamma.weight': hw[n2:n3].clone(), 'proj_out_loggamma.bias': hb[n2:n3].clone(),
            'norm_out.linear.weight': self._view(src, 2).view(2 * D, D).clone(),
            'norm_out.linear.bias': self._view(src, 3).clone(),
        }

    def dropout_seed(self, step_id: int) -> int:
        """Seed of the LoRA dropout masks of one student step: differs per iteration, step and rank (train.py --diff_seed)."""
        return (self.iteration * 7919 + step_id * 104729 + self.reducer.rank * 15485863 + getattr(self, '_chunk', 0) * 32452843 + 12345) & 0x7fffffff

    def _student(self, x, sigma, cond):
        if self.trunk is not None:
            self.trunk.ensure_merged()      # the engine must see W + B A of the LIVE adapters (stale after every optimizer step / checkpoint load)
        return self.student(x.to(torch.bfloat16), sigma, cond['prompt_embeds'], cond.get('pooled'),
                            self._guid(x.shape[0]), cond['hp'], cond['wp'])

    def student_forward(self, x, sigma, cond):
        """The live student through the inference engine (validation, sampling, export checks): with LoRA adapters the engine is re-bound to
        W + B A of the current parameters first (lazily: only when an optimizer step or a checkpoint load changed them since).  Without dropout
        this equals student_forward_unmerged() up to one bf16 rounding of the merged weights."""
        return self._student(x, sigma, cond)

    def _guid(self, B, teacher: bool = False):
        if self.family != 'flux' or not self.student.guidance_embeds:
            return None
        g = self.cfg.guidance
        if teacher and self.cfg.teacher_guidance is not None:
            g = self.cfg.teacher_guidance
        return torch.full((B,), g, device=self.device)

    def _teacher_prepare(self, sigmas, cond, N: int, T: int) -> bool:
        """AdaLN modulation vectors of the teacher's next forwards -- `sigmas` [k, B], k x B <= 8 -- in ONE pass over the stacked modulation matrix
        (``afx_mmdit_prepare_steps``, the inference loop's mechanism: 6.5 GB of weight streaming per FLUX forward, 8 right-hand sides cost what one does).
        The reference re-evaluates `norm1.linear(silu(temb))` in every block of every teacher call (arcflux.py:160-230); the numbers are bit-identical
        (tests/test_hip_engine.py::test_prepared_steps_are_bit_identical_to_the_plain_forward)."""
        B = sigmas.shape[1]
        return self.teacher.prepare_steps(sigmas, cond.get('pooled'), self._guid(B, teacher=True), B, N, T)

    def _teacher_u(self, x, sigma, cond, prepared_step=None):
        """Teacher velocity (GaussianFlow.forward_u, gaussian_flow.py:224-254).  True CFG: the reference runs ONE forward on a 2B batch [negative; positive];
        here the two halves are two forwards of B samples each (per-sample results are identical, the engine's micro-batch holds at most 4 samples).
        prepared_step: this call's row block of the last _teacher_prepare()."""
        xb = x.to(torch.bfloat16)
        g = self._guid(x.shape[0], teacher=True)
        if self.cfg.teacher_guidance_scale != 1.0 and 'negative_prompt_embeds' not in cond:
            raise ValueError('teacher_guidance_scale > 1 (true CFG) needs cond["negative_prompt_embeds"] '
                             '(reference: negative_prompt_embeds_path, lakonlab/datasets/image_prompts.py:158-163)')
        pos = self.teacher(xb, sigma, cond['prompt_embeds'], cond.get('pooled'), g, cond['hp'], cond['wp'], prepared_step=prepared_step).float()
        if self.cfg.teacher_guidance_scale == 1.0:
            return pos
        # the negative pass may reuse the prepared vectors when the conditioning does not see the prompt (Qwen-Image: timestep only; FLUX adds the pooled CLIP vector)
        neg_prep = prepared_step if cond.get('pooled') is None and cond.get('negative_pooled') is None else None
        neg = self.teacher(xb, sigma, cond['negative_prompt_embeds'], cond.get('negative_pooled'), g, cond['hp'], cond['wp'], prepared_step=neg_prep).float()
        return ops.cfg_combine(pos, neg, self.cfg.teacher_guidance_scale)

    def student_forward_unmerged(self, x_src, sigma_src, cond, p_drop: float = 0.0, seed: int = 0):
        """The student's forward as peft evaluates it -- y = W x + B (A dropout(x)) per adapted linear, NOT folded into W -- on the
        LoRA trunk: the engine runs conditioning + embedders (stage 1, with the timestep embedding of the LoRA-adapted embedder
        handed in) and norm_out + head (stage 2); the blocks in between go through the trunk's own block forward, which also keeps
        every block's input (checkpoint) and the pre-gate branch outputs the modulation gradients need.  Returns
        (ArcFlowModelOutput, mod_all [B, n_mod])."""
        B, N, _ = x_src.shape
        T = cond['prompt_embeds'].shape[1]
        dev = self.device
        nb = self.student.num_double + self.student.num_single
        if self._ckpt is None or self._ckpt.shape[1] != B * (T + N):
            self._ckpt = torch.empty(nb, B * (T + N), self.D, dtype=torch.bfloat16, device=dev)
        self.trunk.p_drop = p_drop
        self.trunk.seed = seed
        self.student.set_checkpoint_buffer(None)
        temb_t = self.trunk.temb_forward(sigma_src)
        self.student.set_temb_override(temb_t)
        args = (x_src.to(torch.bfloat16), sigma_src, cond['prompt_embeds'], cond.get('pooled'), self._guid(B), cond['hp'], cond['wp'])
        self.student(*args, stage=1)
        xt = torch.empty(B * (T + N), self.D, dtype=torch.bfloat16, device=dev)
        mod_all = torch.empty(B, self.student.n_mod, dtype=torch.float32, device=dev)
        self.student.export('x_tokens', xt, B, N, T)
        self.student.export('mod_all', mod_all, B, N, T)
        for b in range(B):
            self.trunk.forward_sample(xt, self._ckpt, b, mod_all, T, N, cond['hp'], cond['wp'])
        self.student.import_tokens(xt, B, N, T)
        out = self.student(*args, stage=2)
        self.student.set_temb_override(None)
        return out, mod_all

    # ------------------------------------------------------------------ one student segment
    def _segment(self, step_id: int, x_src, raw_src, cond, teacher_ratio: float, segment: float, rng, draws=None,
                 batch_total: Optional[int] = None, final: bool = False):
        """piid_segment_momentum + the head backward of this student step.  Returns (x_dst, raw_dst).
        batch_total: samples of the whole per-GPU batch (the loss is a mean over it; x_src may be a micro-batch of it).
        final: this is the last gradient-producing call of the iteration -> finished slices of the flat gradient buffer
        are handed to the reducer as the last sample's backward leaves them."""
        c = self.cfg
        B, N, ch = x_src.shape
        if B > 4:
            raise ValueError('a micro-batch holds at most 4 samples (train_step splits larger batches)')
        batch_total = B if batch_total is None else batch_total
        dev = self.device
        K, pp = self.K, self.L
        sigma_src = warp(raw_src, c.shift)
        T = cond['prompt_embeds'].shape[1]
        if self.trunk is not None:
            nb = self.student.num_double + self.student.num_single
            if self._ckpt is None or self._ckpt.shape[1] != B * (T + N):
                self._ckpt = torch.empty(nb, B * (T + N), self.D, dtype=torch.bfloat16, device=dev)
        mod_all = None
        if self.trunk is not None:
            out, mod_all = self.student_forward_unmerged(x_src, sigma_src, cond, c.lora_dropout, self.dropout_seed(step_id))
        else:
            out = self._student(x_src, sigma_src, cond)
        means, logw, logg = out.means, out.logweights, out.loggammas
        xn = torch.empty(B * N, self.D, dtype=torch.bfloat16, device=dev)
        xf = torch.empty(B * N, self.D, dtype=torch.bfloat16, device=dev)
        semb = torch.empty(B, self.D, dtype=torch.float32, device=dev)
        self.student.export('head_in', xn, B, N, T)
        self.student.export('x_final', xf, B, N, T)
        self.student.export('silu_temb', semb, B, N, T)
        if self.trunk is not None:
            temb_sum = torch.empty(B, self.D, dtype=torch.float32, device=dev)       # pre-SiLU conditioning sum: d silu for the embedder pair
            self.student.export('temb', temb_sum, B, N, T)
        if self.trunk is not None and mod_all is None:
            mod_all = torch.empty(B, self.student.n_mod, dtype=torch.float32, device=dev)
            self.student.export('mod_all', mod_all, B, N, T)

        n_sub = max(round(segment * c.total_substeps), 1)
        window = min(c.window_substeps * (segment / n_sub), segment)
        raw_dst = raw_src - segment
        # GM dropout of the roll-out policy (policies/arcflow.py:96-106)
        n = c.num_intermediate_states
        if draws is None:      # same draw order as the reference: dropout, student intervals, teacher intervals
            draws = (torch.rand(B, K, device=dev, generator=rng), torch.rand(B, n, device=dev, generator=rng),
                     torch.rand(B, n - 1, device=dev, generator=rng))
        u_drop, u_stu, u_tea = (d.to(dev) for d in draws)
        drop = u_drop < c.gm_dropout
        drop &= ~drop.all(dim=1, keepdim=True)
        if c.gm_dropout <= 0:
            drop = None
        span = segment - window
        s_iv = torch.sort(u_stu * ((1 - teacher_ratio) * span), dim=-1)[0]
        s_iv = torch.diff(s_iv, dim=-1, prepend=torch.zeros(B, 1, device=dev))
        t_iv = torch.sort(u_tea, dim=-1)[0]
        t_iv = torch.diff(t_iv, dim=-1, prepend=torch.zeros(B, 1, device=dev), append=torch.ones(B, 1, device=dev)) \
            * (teacher_ratio * span)

        d_means = torch.zeros(B, N, K, ch, dtype=torch.float32, device=dev)
        d_logw = torch.zeros(B, N, K, pp, dtype=torch.float32, device=dev)
        d_logg = torch.zeros(B, N, K - 1, pp, dtype=torch.float32, device=dev)
        coef = c.loss_scale / (n * batch_total * N * ch) * segment          # mean over the 4B stacked states x segment weight
        x, raw, sigma = x_src, raw_src, sigma_src
        one = torch.ones(B, device=dev)
        # the teacher's timesteps depend on the interval draws only, not on the roll-out: the modulation vectors of the next `chunk` teacher states come out of
        # one pass over the stacked modulation matrix instead of one pass per forward (ARCFLOW_TRAIN_PREP_MOD=0: per forward, for A/B runs)
        sig_a_all, raw_t = [], raw_src
        for i in range(n):
            raw_t = (raw_t - s_iv[:, i]).clamp(min=0)
            sig_a_all.append(warp(raw_t, c.shift))
            raw_t = (raw_t - t_iv[:, i]).clamp(min=0)
        chunk = min(max(8 // B, 1), n) if os.environ.get('ARCFLOW_TRAIN_PREP_MOD', '1') != '0' else 0
        prepared = False
        for i in range(n):
            raw_a = (raw - s_iv[:, i]).clamp(min=0)
            raw_b = (raw_a - t_iv[:, i]).clamp(min=0)
            sigma_a = warp(raw_a, c.shift)
            x_a = ops.arcflow_step_dropout(x, means, logw, logg, sigma_src, sigma, sigma_a, drop, c.eps)
            if chunk > 1 and i % chunk == 0:
                prepared = self._teacher_prepare(torch.stack(sig_a_all[i:i + chunk]), cond, N, T)
            tgt = self._teacher_u(x_a, sigma_a, cond, prepared_step=(i % chunk) if (chunk > 1 and prepared) else None)
            # predicted mean velocity of the full policy over [raw_e, raw_a] (policy_average_u_momentum)
            raw_e = raw_b - window
            short = (torch.round((raw_a - raw_e) * c.total_substeps) < 2).float()
            sigma_e = warp(raw_e, c.shift)
            inv_den = 1.0 / (sigma_a - sigma_e).clamp(min=c.eps)
            x_e = ops.arcflow_step(x_a, means, logw, logg, sigma_src, sigma_a, sigma_e, c.eps)
            u_mean = ops.axpby_rows(x_a, inv_den, x_e, -inv_den)
            u_loc = ops.arcflow_velocity(means, logw, logg, sigma_src, sigma_a)
            pred = ops.axpby_rows(u_loc, short, u_mean, one - short)
            g = ops.mse_loss(pred, tgt, coef, self._loss_acc)
            ops.arcflow_backward(g, means, logw, logg, sigma_src, sigma_a, sigma_e, gscale=(one - short) * inv_den,
                                 grads=(d_means, d_logw, d_logg), eps=c.eps)
            ops.arcflow_backward(g, means, logw, logg, sigma_src, sigma_a, sigma_a, gscale=short, velocity=True,
                                 grads=(d_means, d_logw, d_logg), eps=c.eps)
            sigma_b = warp(raw_b, c.shift)
            x = ops.euler_roll(x_a, tgt, sigma_a, sigma_b)
            raw, sigma = raw_b, sigma_b
        x_dst = ops.arcflow_step_dropout(x, means, logw, logg, sigma_src, sigma, warp(raw_dst, c.shift), drop, c.eps)

        # ---- backward: head logits -> head weights / bias, norm_out modulation -> norm_out.linear ---------------
        gbuf = self.grad
        dy = ops.head_grad(d_means, d_logw, d_logg, logw, self.head_n)                 # [M, head_n] bf16
        M = B * N
        gw = self._view(gbuf, 0).view(self.head_n, self.D)
        if self.head_n % 8 == 0 and gw.data_ptr() % 16 == 0:     # contraction over the tokens, straight from the token-major operands (afx_tn.hip)
            ops.linear_tn_f32out(dy[:, :self.head_n], xn, out=gw, accumulate=True)
        else:
            if M % 64:
                raise ValueError('batch x tokens must be a multiple of 64 for the weight-gradient GEMM')
            ops.linear_f32out(ops.transpose(dy), ops.transpose(xn), out=gw, accumulate=True)
        ops.colsum(dy, self._view(gbuf, 1))
        dxn = ops.linear(dy, ops.transpose(self.w_head))                               # [M, D] bf16 = dY . W_head
        dmod = ops.normout_backward(xf, dxn, torch.zeros(B, 2, self.D, dtype=torch.float32, device=dev), N)
        dflat = dmod.view(B, 2 * self.D)
        ops.outer_accum(dflat, semb, self._view(gbuf, 2).view(2 * self.D, self.D))
        ops.outer_accum(dflat, torch.ones(B, 1, device=dev), self._view(gbuf, 3).view(2 * self.D, 1))
        if self.trunk is not None:                          # LoRA adapters: per-sample recompute + backward of every block
            dmod_all = torch.zeros(B, self.student.n_mod, dtype=torch.float32, device=dev)
            for b in range(B):
                done = self._launch_block_slice if (final and b == B - 1) else None
                self.trunk.backward_sample(self._ckpt, b, mod_all, xf[b * N:(b + 1) * N], dxn[b * N:(b + 1) * N], T, N,
                                           cond['hp'], cond['wp'], gbuf, dmod_out=dmod_all[b], on_block_done=done)
            # timestep-embedder LoRA pair: d silu(temb) = W_mod^T d mod (all blocks) + W_norm_out^T d mod_final, then SiLU'
            dsemb = torch.zeros(B, self.D, dtype=torch.float32, device=dev)
            ops.gemv_t(dmod_all, self.student._weights['mod.weight'], dsemb)
            ops.gemv_t(dflat, self.w_no, dsemb)                 # the bf16 working copy the forward multiplied with
            sg = torch.sigmoid(temb_sum)
            self.trunk.temb_backward(dsemb * (sg * (1 + temb_sum * (1 - sg))), gbuf)
        return x_dst, raw_dst

    # ------------------------------------------------------------------ gradient exchange, slice by slice
    def _launch_block_slice(self, block: int) -> None:
        """Called by the trunk when ``block``'s backward of the iteration's LAST sample is done: its adapters' gradients are
        final -> start their all-reduce now (it overlaps the remaining blocks' backward)."""
        a, b = self.trunk.block_slice(block)
        self.reducer.launch(self.grad[a:b])
        self._launched.append((a, b))

    def _launch_rest(self) -> None:
        """Everything not exchanged yet (heads, norm_out, the timestep-embedder pair; the whole buffer without a trunk)."""
        pos = 0
        for a, b in sorted(self._launched):
            if a > pos:
                self.reducer.launch(self.grad[pos:a])
            pos = max(pos, b)
        if pos < self.grad.numel():
            self.reducer.launch(self.grad[pos:])
        self._launched = []

    # ------------------------------------------------------------------ one iteration
    def lr_at(self, it: int) -> float:
        c = self.cfg
        if it < c.warmup_iters:
            k = (1 - it / c.warmup_iters) * (1 - c.warmup_ratio)
            return c.lr * (1 - k)
        return c.lr

    def train_step(self, cond: dict, batch: int, rng: Optional[torch.Generator] = None, x_init: Optional[torch.Tensor] = None,
                   draws=None):
        """cond: prompt_embeds [B,T,joint] (+ pooled, negative_*), hp, wp.  Returns a dict of python floats
        (loss, grad_norm, lr, teacher_ratio, skipped) -- one host sync per step, like the reference's float(loss)."""
        c = self.cfg
        N = cond['hp'] * cond['wp']
        it = self.iteration
        teacher_ratio = 1 - min(it, c.num_decay_iters) / c.num_decay_iters if c.num_decay_iters > 0 else 0.0
        self.grad.zero_()
        self._loss_acc.zero_()
        self._launched = []
        x_all = x_init if x_init is not None else torch.randn(batch, N, self.C, device=self.device, generator=rng)
        base = 1.0 / (c.nfe - 1 + max(c.timestep_ratio, c.eps))
        chunks = [(a, min(a + 4, batch)) for a in range(0, batch, 4)]      # micro-batches of <= 4 samples
        outs = []
        for ci, (a, b) in enumerate(chunks):
            self._chunk = ci                                   # part of the LoRA-dropout seed: micro-batches draw different masks
            x = x_all[a:b]
            raw = torch.ones(b - a, device=self.device)
            cc = cond if len(chunks) == 1 else {k: (v[a:b] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == batch else v)
                                                for k, v in cond.items()}
            for step_id in range(c.nfe):
                seg = base * max(c.timestep_ratio, c.eps) if step_id == c.nfe - 1 else base
                dr = None if draws is None else tuple(d[a:b] for d in draws[step_id])
                x, raw = self._segment(step_id, x, raw, cc, teacher_ratio, seg, rng, dr, batch_total=batch,
                                       final=(ci == len(chunks) - 1 and step_id == c.nfe - 1))
            outs.append(x)
        x = outs[0] if len(outs) == 1 else torch.cat(outs)
        self._launch_rest()                                    # heads / norm_out / embedder pair (or everything)
        inv_world = self.reducer.finish()
        g = self.grad
        self._norm_acc.zero_()
        ops.sumsq(g, self._norm_acc)
        grad_norm = float(self._norm_acc.sqrt()) * inv_world   # the step's one host sync
        loss = float(self._loss_acc)
        lr = self.lr_at(it)
        skipped = not (grad_norm == grad_norm and grad_norm != float('inf'))
        if not skipped:
            scale = inv_world
            if it >= c.grad_clip_begin_iter and c.grad_clip > 0 and grad_norm > c.grad_clip:
                scale *= c.grad_clip / (grad_norm + 1e-6)
            self.opt_steps += 1
            groups = [(a, b, lr * (c.loggamma_lr_mult if lg else 1.0)) for a, b, lg in self.optimizer_groups()]
            for a, b, glr in groups:
                if b <= a:
                    continue
                if c.optimizer == 'adamw8bit' and b - a >= 4096:       # bitsandbytes min_8bit_size
                    if (a, b) not in self.opt8:
                        self.opt8[(a, b)] = ops.AdamW8bitState(b - a, self.device)
                    ops.adamw8bit_step(self.params[a:b], g[a:b], self.opt8[(a, b)], glr, self.opt_steps, betas=c.betas,
                                       weight_decay=c.weight_decay, grad_scale=scale)
                    continue
                if c.optimizer == 'adamw8bit':
                    if (a, b) not in self._small:
                        self._small[(a, b)] = (torch.zeros(b - a, device=self.device), torch.zeros(b - a, device=self.device))
                    m_, v_ = self._small[(a, b)]
                else:
                    m_, v_ = self.exp_avg[a:b], self.exp_avg_sq[a:b]
                ops.adamw_step(self.params[a:b], g[a:b], m_, v_, glr, self.opt_steps,
                               betas=c.betas, weight_decay=c.weight_decay, grad_scale=scale)
            self._sync_working_copies()
            if self.trunk is not None:
                self.trunk.refresh()
        # Karras EMA (ema_hook.py:86-124): copy before start_iter, lerp after
        if it < c.ema_start_iter:
            self.ema.copy_(self.params)
        else:
            t = max(it + 1 - c.ema_start_iter, 1)
            ops.ema_lerp(self.ema, self.params, min((1 - 1 / t) ** (c.ema_gamma + 1), 1.0))
        self.iteration += 1
        self.last_x = x
        return dict(loss=loss, grad_norm=grad_norm, lr=lr, teacher_ratio=teacher_ratio, skipped=skipped,
                    allreduce_exposed_ms=self.reducer.exposed_ms())
