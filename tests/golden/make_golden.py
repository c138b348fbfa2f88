#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by EXECUTING the reference's own functions.

Runs only in the build container (needs /root/reference); the GPU box never sees the
reference.  Nothing from the reference is copied into this repository: its source files are
parsed in place, the pure functions on the ArcFlow hot path are pulled out of the AST and
executed with stub ``self`` objects, and only the resulting tensors (inputs + outputs) are
written to ``tests/golden/*.npz``.

Functions executed (paths under /root/reference/lakonlab):
  pipelines/arcflux_pipeline.py : retrieve_raw_timesteps, ArcFluxPipeline.{momentum_integration,
                                  _unpack_mp,_pack_latents,_unpack_latents}
  pipelines/arcqwen_pipeline.py : ArcQwenImagePipeline.momentum_integration (return_mid)
  models/diffusions/policies/{base,arcflow}.py : ArcFlowPolicy (imported as a module)
  models/diffusions/sampler.py  : ContinuousTimeStepSampler (class body, decorator stripped)
  models/diffusions/arcflow.py  : ArcFlowImitationBase.{momentum_integration,
                                  policy_average_u_momentum,piid_segment_momentum,get_shape_info}
  models/diffusions/gaussian_flow.py : guidance_jit (decorator stripped)
  models/architecture/arcflow/arcflux.py : ArcFluxTransformer2DModel.{patchify,unpatchify},
                                  _ArcFluxTransformer2DModel.init_weights (log-gamma bias only)
  runner/hooks/ema_hook.py      : ExponentialMovingAverageHookMod.karras

Usage:  python tests/golden/make_golden.py        (rewrites every fixture deterministically)
"""
import ast
import contextlib
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/lakonlab'
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- loaders
def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_policy():
    pkg = types.ModuleType('refpol')
    pkg.__path__ = [REF + '/models/diffusions/policies']
    sys.modules['refpol'] = pkg
    load_module('refpol.base', REF + '/models/diffusions/policies/base.py')
    return load_module('refpol.arcflow', REF + '/models/diffusions/policies/arcflow.py').ArcFlowPolicy


def grab(path, names, cls=None, ns=None):
    """exec the FunctionDefs ``names`` (module level, or inside class ``cls``) of ``path``."""
    ns = ns if ns is not None else {}
    tree = ast.parse(open(path).read())
    body = tree.body
    if cls is not None:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    found = set()
    for node in body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            mod = ast.Module([node], [])
            ast.fix_missing_locations(mod)
            exec(compile(mod, path, 'exec'), ns)
            found.add(node.name)
    missing = set(names) - found
    assert not missing, f'{path}: missing {missing}'
    return ns


def grab_class(path, cls, ns):
    tree = ast.parse(open(path).read())
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
    node.decorator_list = []
    mod = ast.Module([node], [])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, 'exec'), ns)
    return ns[cls]


class Obj:
    pass


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'wrote {name}.npz  ({", ".join(f"{k}{list(v.shape)}" for k, v in out.items())})')


def rand_mixture(gen, b, k, c, h, w, logg_scale=1.0):
    means = torch.randn(b, k, c, h, w, generator=gen)
    logw = torch.log_softmax(torch.randn(b, k, 1, h, w, generator=gen) * 2.0, dim=1)
    logg = torch.randn(b, k - 1, 1, h, w, generator=gen) * logg_scale
    return means, logw, logg


G9_STRIDE = 61            # co-prime with 64 channels and 4096 tokens: the sample walks every channel and patch position


def golden_full_size_inputs(seed=20260929):
    """The seeded full-size (4096 tokens, K = 16) mixture + latent of fixture G9, in the token layout the denoiser emits."""
    gen = torch.Generator().manual_seed(seed)
    n = 4096
    m_tok = torch.randn(1, n, 16, 64, generator=gen)
    lw_tok = torch.log_softmax(torch.randn(1, n, 16, 4, generator=gen) * 2.0, dim=-2)
    lg_tok = torch.randn(1, n, 15, 4, generator=gen) * 1.2
    lg_tok[0, :64, 0] = 0.0                  # the clamp / sign branches of phi at full size too
    lg_tok[0, 64:128, 1] = 1e-5
    lg_tok[0, 128:192, 2] = -1e-5
    x_tok = torch.randn(1, n, 64, generator=gen)
    return dict(means_tok=m_tok, logw_tok=lw_tok, logg_tok=lg_tok, x_tok=x_tok)


def main():
    torch.set_num_threads(4)
    ArcFlowPolicy = load_policy()
    base_ns = {'torch': torch, 'np': np, 'math': math}

    flux = grab(REF + '/pipelines/arcflux_pipeline.py', ['retrieve_raw_timesteps'], ns=dict(base_ns))
    grab(REF + '/pipelines/arcflux_pipeline.py',
         ['momentum_integration', '_unpack_mp', '_pack_latents', '_unpack_latents'],
         cls='ArcFluxPipeline', ns=flux)
    qwen = grab(REF + '/pipelines/arcqwen_pipeline.py', ['momentum_integration'],
                cls='ArcQwenImagePipeline', ns=dict(base_ns))

    pipe = Obj()
    pipe.scheduler = Obj(); pipe.scheduler.config = Obj()
    pipe.scheduler.config.num_train_timesteps = 1000
    pipe.num_timesteps = 128
    pipe.vae_scale_factor = 8
    pipe.transformer = Obj(); pipe.transformer.num_gaussians = 16

    # ---------------------------------------------------------------- G1 time grid
    g1 = {}
    for nfe, ratio in [(2, 1.0), (4, 1.0), (4, 0.5), (1, 1.0), (3, 0.25), (8, 1.0)]:
        raw, counts, total = flux['retrieve_raw_timesteps'](nfe, 128, ratio)
        tag = f'n{nfe}_r{str(ratio).replace(".", "p")}'
        g1[tag + '_raw'] = np.asarray(raw, dtype=np.float64)
        g1[tag + '_counts'] = np.asarray(counts, dtype=np.int64)
        g1[tag + '_total'] = np.asarray(total, dtype=np.int64)
    save('g1_time_grid', **g1)

    # ---------------------------------------------------------------- G2 pipeline-form step
    gen = torch.Generator().manual_seed(1234)
    b, k, c, h, w = 2, 16, 16, 8, 8
    means, logw, logg = rand_mixture(gen, b, k, c, h, w, logg_scale=1.2)
    # force the clamp / sign branches of phi: exact zeros, tiny +-, negative rates
    logg[0, 0, 0, 0, :] = 0.0
    logg[0, 1, 0, 0, :] = 1e-5
    logg[0, 2, 0, 0, :] = -1e-5
    logg[1, 3] = -torch.abs(logg[1, 3])
    x = torch.randn(b, c, h, w, generator=gen)
    g2 = dict(means=means, logw=logw, logg=logg, x=x)
    cases = [(1.0, 1000 * 0.7619047761), (0.7619047761, 0.0), (0.5, 250.0)]
    for i, (s_src, t_end) in enumerate(cases):
        pol = ArcFlowPolicy(dict(means=means.clone(), logweights=logw.clone(), loggammas=logg.clone()),
                            x, torch.tensor(s_src))
        x_end, s_end, t_e = flux['momentum_integration'](
            pipe, torch.tensor(s_src), x, torch.tensor(s_src), torch.tensor(t_end), pol, eps=1e-4)
        g2[f'case{i}_sigma_src'] = np.float32(s_src)
        g2[f'case{i}_t_end'] = np.float32(t_end)
        g2[f'case{i}_x_end'] = x_end
        g2[f'case{i}_sigma_end'] = s_end
        g2[f'case{i}_x0_means'] = pol.denoising_output_x_0['means']
    # Qwen variant with return_mid
    pol = ArcFlowPolicy(dict(means=means.clone(), logweights=logw.clone(), loggammas=logg.clone()),
                        x, torch.tensor(1.0))
    r = qwen['momentum_integration'](pipe, torch.tensor(1.0), x, torch.tensor(1.0),
                                     torch.tensor(761.9047761), pol, eps=1e-4, return_mid=True)
    g2['qwen_x_end'] = r[0]
    g2['qwen_x_mid'] = r[-1] if len(r) > 3 else r[0]
    g2['qwen_num_returns'] = np.int64(len(r))
    save('g2_step_pipeline', **g2)

    # ---------------------------------------------------------------- G5 layouts (+ a packed full step)
    gen = torch.Generator().manual_seed(99)
    bsz, hh, ww = 2, 6, 10                      # latent 12 x 20 -> tokens 6 x 10
    height, width = hh * 16, ww * 16            # pixel size seen by the pipeline helpers
    lat = torch.arange(bsz * 16 * 2 * hh * 2 * ww, dtype=torch.float32).reshape(bsz, 16, 2 * hh, 2 * ww)
    packed = flux['_pack_latents'](lat, bsz, 16, 2 * hh, 2 * ww, patch_size=1)
    unpacked = flux['_unpack_latents'](packed, height, width, 8, target_patch_size=1)
    m_tok = torch.randn(bsz, hh * ww, 16, 64, generator=gen)
    lw_tok = torch.log_softmax(torch.randn(bsz, hh * ww, 16, 4, generator=gen), dim=-2)
    lg_tok = torch.randn(bsz, hh * ww, 15, 4, generator=gen)
    mp = flux['_unpack_mp'](pipe, dict(means=m_tok.clone(), logweights=lw_tok.clone(), loggammas=lg_tok.clone()),
                            height, width, 16, gm_patch_size=1)
    # training-side twins
    arc = grab(REF + '/models/architecture/arcflow/arcflux.py', ['patchify', 'unpatchify'],
               cls='ArcFluxTransformer2DModel', ns=dict(base_ns))
    mdl = Obj(); mdl.patch_size = 2
    pat = arc['patchify'](mdl, lat)
    tr_means = m_tok.permute(0, 2, 3, 1).reshape(bsz, 16, 64, hh, ww)
    tr_lw = lw_tok.permute(0, 2, 3, 1).reshape(bsz, 16, 4, hh, ww)
    tr_lg = lg_tok.permute(0, 2, 3, 1).reshape(bsz, 15, 4, hh, ww)
    unp = arc['unpatchify'](mdl, dict(means=tr_means.clone(), logweights=tr_lw.clone(), loggammas=tr_lg.clone()))
    # full pipeline step in token layout: unpack -> policy -> integrate -> repack
    x_tok = torch.randn(bsz, hh * ww, 64, generator=gen)
    x_lat = flux['_unpack_latents'](x_tok, height, width, 8, target_patch_size=1)
    pol = ArcFlowPolicy({k2: v.to(torch.float32) for k2, v in mp.items()}, x_lat, torch.tensor(1.0))
    x_end = flux['momentum_integration'](pipe, torch.tensor(1.0), x_lat, torch.tensor(1.0),
                                         torch.tensor(761.9047761), pol, eps=1e-4)[0]
    x_end_tok = flux['_pack_latents'](x_end, bsz, 16, 2 * hh, 2 * ww, patch_size=1)
    save('g5_layouts', lat=lat, packed=packed, unpacked=unpacked,
         means_tok=m_tok, logw_tok=lw_tok, logg_tok=lg_tok,
         means_lat=mp['means'], logw_lat=mp['logweights'], logg_lat=mp['loggammas'],
         patchified=pat, unp_means=unp['means'], unp_logw=unp['logweights'], unp_logg=unp['loggammas'],
         x_tok=x_tok, x_end_tok=x_end_tok, hp=np.int64(hh), wp=np.int64(ww))

    # ---------------------------------------------------------------- G9 one FULL-SIZE step (1024^2: latent [1, 16, 128, 128])
    # Inputs are a seeded CPU draw the test repeats (their fp64 checksums are stored so a changed generator is noticed); of the
    # 262 144 outputs the fixture keeps a strided sample + fp64 moments, for both steps of the 2-NFE schedule.
    g9 = golden_full_size_inputs()
    hh = ww = 64
    height = width = hh * 16
    mp = flux['_unpack_mp'](pipe, dict(means=g9['means_tok'].clone(), logweights=g9['logw_tok'].clone(), loggammas=g9['logg_tok'].clone()),
                            height, width, 16, gm_patch_size=1)
    x_lat = flux['_unpack_latents'](g9['x_tok'], height, width, 8, target_patch_size=1)
    assert tuple(x_lat.shape) == (1, 16, 128, 128)
    out9 = dict(stride=np.int64(G9_STRIDE), **{'in_sum_' + k2: np.float64(v.double().sum().item()) for k2, v in g9.items()})
    for i, (s_src, t_end) in enumerate([(1.0, 761.9047761), (0.7619047761, 0.0)]):
        pol = ArcFlowPolicy({k2: v.to(torch.float32) for k2, v in mp.items()}, x_lat, torch.tensor(s_src))
        x_end = flux['momentum_integration'](pipe, torch.tensor(s_src), x_lat, torch.tensor(s_src), torch.tensor(t_end), pol, eps=1e-4)[0]
        x_end_tok = flux['_pack_latents'](x_end, 1, 16, 2 * hh, 2 * ww, patch_size=1)
        flat = x_end_tok.reshape(-1)
        out9[f'case{i}_sigma_src'] = np.float32(s_src)
        out9[f'case{i}_t_end'] = np.float32(t_end)
        out9[f'case{i}_sample'] = flat[::G9_STRIDE].clone()
        out9[f'case{i}_sum'] = np.float64(flat.double().sum().item())
        out9[f'case{i}_sumsq'] = np.float64((flat.double() ** 2).sum().item())
        out9[f'case{i}_absmax'] = np.float32(flat.abs().max().item())
    save('g9_step_full_size', **out9)

    # ---------------------------------------------------------------- training-form pieces
    samp_ns = dict(base_ns)
    Sampler = grab_class(REF + '/models/diffusions/sampler.py', 'ContinuousTimeStepSampler', samp_ns)
    trn = dict(base_ns)
    trn['ArcFlowPolicy'] = ArcFlowPolicy
    trn['module_eval'] = lambda m: contextlib.nullcontext()
    grab(REF + '/models/diffusions/arcflow.py',
         ['momentum_integration', 'policy_average_u_momentum', 'piid_segment_momentum', 'get_shape_info'],
         cls='ArcFlowImitationBase', ns=trn)
    dif = Obj()
    dif.timestep_sampler = Sampler(num_timesteps=1, shift=3.2)
    dif.num_timesteps = 1
    dif.momentum_integration = types.MethodType(trn['momentum_integration'], dif)
    dif.policy_average_u_momentum = types.MethodType(trn['policy_average_u_momentum'], dif)
    dif.get_shape_info = trn['get_shape_info']
    warp = dif.timestep_sampler.warp_t

    # G3: training-form integration, sigma_start != sigma_src, per-sample end times
    gen = torch.Generator().manual_seed(77)
    b, k, c, h, w = 3, 16, 16, 6, 6
    means, logw, logg = rand_mixture(gen, b, k, c, h, w)
    x = torch.randn(b, c, h, w, generator=gen)
    raw_src = torch.tensor([1.0, 0.5, 1.0])
    raw_a = torch.tensor([0.9, 0.37, 0.61])
    raw_end = torch.tensor([0.55, 0.0, 0.5])
    s_src = warp(raw_src).reshape(b, 1, 1, 1)
    s_a = warp(raw_a).reshape(b, 1, 1, 1)
    pol = ArcFlowPolicy(dict(means=means, logweights=logw, loggammas=logg), x, s_src)
    x_end, s_end, t_end = dif.momentum_integration(s_src, x, s_a, raw_end, pol, eps=1e-4)
    save('g3_step_training', means=means, logw=logw, logg=logg, x=x, raw_src=raw_src, raw_a=raw_a,
         raw_end=raw_end, sigma_src=s_src, sigma_a=s_a, x_end=x_end, sigma_end=s_end, t_end=t_end,
         warp_in=torch.linspace(0, 1, 33), warp_out=warp(torch.linspace(0, 1, 33)),
         unwarp_out=dif.timestep_sampler.unwarp_t(torch.linspace(0, 1, 33)))

    # G4: velocity() and policy_average_u_momentum, both branches mixed
    vel = pol.velocity(s_src, s_a)
    raw_b = torch.tensor([0.9 - 1.0 / 128, 0.2, 0.61 - 0.4 / 128])   # sample 0,2 short; sample 1 long
    pred = dif.policy_average_u_momentum(s_src, x, s_a, raw_a, raw_b, 128, pol, eps=1e-4)
    raw_b_long = torch.tensor([0.5, 0.1, 0.3])
    pred_long = dif.policy_average_u_momentum(s_src, x, s_a, raw_a, raw_b_long, 128, pol, eps=1e-4)
    raw_b_short = raw_a - 0.5 / 128
    pred_short = dif.policy_average_u_momentum(s_src, x, s_a, raw_a, raw_b_short, 128, pol, eps=1e-4)
    save('g4_velocity', means=means, logw=logw, logg=logg, x=x, sigma_src=s_src, sigma_a=s_a, raw_a=raw_a,
         velocity=vel, raw_b=raw_b, pred=pred, raw_b_long=raw_b_long, pred_long=pred_long,
         raw_b_short=raw_b_short, pred_short=pred_short)

    # G6: GM dropout + CFG bias + Karras EMA + log-gamma bias init
    torch.manual_seed(4321)
    pol2 = ArcFlowPolicy(dict(means=means, logweights=logw.clone(), loggammas=logg), x, s_src)
    pol2.dropout_(0.1)
    torch.manual_seed(4321)
    u01 = torch.rand(b, k, 1, 1, 1)
    torch.manual_seed(5)
    polh = ArcFlowPolicy(dict(means=means, logweights=logw.clone(), loggammas=logg), x, s_src)
    polh.dropout_(0.97)                                     # exercises the all-dropped rescue
    torch.manual_seed(5)
    u97 = torch.rand(b, k, 1, 1, 1)
    gf = grab(REF + '/models/diffusions/gaussian_flow.py', ['guidance_jit'], ns=dict(base_ns))
    pos = torch.randn(2, 16, 6, 6, generator=gen)
    neg = torch.randn(2, 16, 6, 6, generator=gen)
    ema = grab(REF + '/runner/hooks/ema_hook.py', ['karras'], cls='ExponentialMovingAverageHookMod',
               ns=dict(base_ns))
    hook = Obj(); hook.start_iter = 100
    steps = np.array([0, 50, 99, 100, 101, 102, 110, 201, 1100, 10000])
    betas = []
    for it in steps:
        runner = Obj(); runner.iter = int(it)
        betas.append(ema['karras'](hook, runner, gamma=7.0)['momentum'])
    betas = np.array(betas, dtype=np.float64)
    iw = dict(base_ns)
    iw['constant_init'] = lambda m, val=0: None
    grab(REF + '/models/architecture/arcflow/arcflux.py', ['init_weights'],
         cls='_ArcFluxTransformer2DModel', ns=iw)
    net = Obj()
    net.num_gaussians, net.num_gammas, net.logweights_channels, net.out_channels = 16, 15, 4, 64
    for nm in ('proj_out_means', 'proj_out_logweights', 'proj_out_loggamma'):
        lin = Obj(); lin.to_empty = (lambda self_=lin, device=None: self_)
        lin.bias = Obj()
        lin.bias.data = torch.zeros({'proj_out_means': 1024, 'proj_out_logweights': 64,
                                     'proj_out_loggamma': 60}[nm])
        setattr(net, nm, lin)
    iw['init_weights'](net)
    save('g6_misc', logw=logw, u01=u01, dropped01=torch.isinf(pol2.denoising_output_x_0['logweights']),
         u97=u97, dropped97=torch.isinf(polh.denoising_output_x_0['logweights']),
         pos=pos, neg=neg, cfg_plain=gf['guidance_jit'](pos, neg, 4.0, False),
         cfg_orth=gf['guidance_jit'](pos, neg, 4.0, True),
         ema_steps=steps, ema_betas=betas, loggamma_bias=net.proj_out_loggamma.bias.data)

    # G7: one full distillation segment with an analytic fake teacher
    captured = {}

    def flow_loss(kw):
        captured.update({k2: v.detach().clone() for k2, v in kw.items()})
        return ((kw['u_t_pred'] - kw['u_t']) ** 2).flatten(1).mean(dim=1).mul(0.5 * 30.0).mean()

    def teacher(return_u=True, x_t=None, t=None, **kw):
        return 0.3 * x_t - 0.7 * t.reshape(-1, 1, 1, 1) + 0.05 * torch.roll(x_t, 1, dims=-1)

    for tag, ratio, seg, raw0 in [('a', 0.6, 0.5, 1.0), ('b', 0.0, 0.5, 0.5), ('c', 1.0, 0.25, 0.75)]:
        dif.train_cfg = dict(eps=1e-4, total_substeps=128, num_intermediate_states=4,
                             window_substeps=3, gm_dropout=0.1)
        dif.flow_loss = flow_loss
        dif.piid = types.MethodType(trn['piid_segment_momentum'], dif)
        raw_src = torch.full((b,), raw0)
        s_src = warp(raw_src).reshape(b, 1, 1, 1)
        pol = ArcFlowPolicy(dict(means=means, logweights=logw.clone(), loggammas=logg), x, s_src)
        seed = {'a': 11, 'b': 12, 'c': 13}[tag]
        torch.manual_seed(seed)
        loss, x_dst, raw_dst = dif.piid(teacher, pol, x, raw_src, s_src, ratio, seg, dict(), get_x_t_dst=True)
        torch.manual_seed(seed)
        u_drop = torch.rand(b, k, 1, 1, 1)
        u_stu = torch.rand(b, 4)
        u_tea = torch.rand(b, 3)
        save(f'g7_segment_{tag}', means=means, logw=logw, logg=logg, x=x, raw_src=raw_src,
             teacher_ratio=np.float32(ratio), segment=np.float32(seg), u_drop=u_drop, u_student=u_stu,
             u_teacher=u_tea, loss=loss, x_dst=x_dst, raw_dst=raw_dst,
             u_pred=captured['u_t_pred'], u_tgt=captured['u_t'], timesteps=captured['timesteps'])

    golden_sampler()


def golden_sampler():
    """G8: rank-strided / bucketed / resumable sampler (lakonlab/datasets/samplers/distributed_sampler.py:19-158).
    The class needs only torch + numpy; mmgen's ``sync_random_seed`` (an all-rank broadcast) is stubbed."""
    src = open(os.path.join(REF, 'datasets/samplers/distributed_sampler.py')).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if not (isinstance(n, ast.ImportFrom) and n.module and n.module.startswith('mmgen'))]
    ns = {'sync_random_seed': lambda s: 0 if s is None else s}
    exec(compile(ast.Module(body=body, type_ignores=[]), 'ref_sampler', 'exec'), ns)
    Ref = ns['DistributedSampler']

    class DS:
        def __init__(self, n, buckets=None):
            self.n = n
            if buckets is not None:
                self.bucket_ids = buckets

        def __len__(self):
            return self.n

    out = {}
    g = torch.Generator().manual_seed(0)
    b37 = torch.randint(0, 3, (37,), generator=g).tolist()
    b64 = [0] * 20 + [1] * 31 + [2] * 13
    cases = [(23, None, 2, 4, True, 7, 0, 0), (23, None, 2, 4, True, 7, 1, 2), (40, None, 8, 2, False, 0, 0, 1),
             (64, b64, 2, 4, True, 3, 0, 0), (64, b64, 4, 2, True, 3, 2, 3), (37, b37, 2, 2, False, 0, 0, 0),
             (64, b64, 8, 4, True, 11, 0, 0)]
    for ci, (n, buckets, world, spg, shuffle, seed, epoch, it) in enumerate(cases):
        for rank in range(world):
            s = Ref(DS(n, buckets), num_replicas=world, rank=rank, shuffle=shuffle, samples_per_gpu=spg, seed=seed)
            s.set_epoch(epoch)
            s.set_iter(it)
            out[f'c{ci}_r{rank}'] = np.array(list(iter(s)), dtype=np.int64)
        out[f'c{ci}_meta'] = np.array([n, world, spg, int(shuffle), seed, epoch, it, -1 if buckets is None else 1], dtype=np.int64)
        if buckets is not None:
            out[f'c{ci}_buckets'] = np.array(buckets, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, 'g8_sampler.npz'), **out)


if __name__ == '__main__':
    main()
