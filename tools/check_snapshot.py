#!/usr/bin/env python3
"""Validate a local diffusers snapshot (FLUX.1-dev / Qwen-Image) or an ArcFlow adapter directory against what the
MI355X engine expects -- safetensors HEADERS only, nothing is loaded (SURVEY 8c self-check 1).

    python tools/check_snapshot.py <snapshot dir> [--family flux|qwen] [--adapter <adapter dir>]

Exit code 0 when every expected tensor is present with the expected shape, 1 otherwise (missing / mis-shaped tensors listed).
The reference consumes the same names through diffusers' from_pretrained and lakonlab/pipelines/arcflow_loader.py:241-263."""
import argparse
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def header_shapes(folder):
    from safetensors import safe_open
    shapes = {}
    files = sorted(glob.glob(os.path.join(folder, '*.safetensors')))
    if not files:
        raise SystemExit(f'{folder}: no *.safetensors file')
    idx = os.path.join(folder, 'diffusion_pytorch_model.safetensors.index.json')
    if os.path.exists(idx):                       # sharded checkpoint: the index must name every shard that exists and vice versa
        wm = json.load(open(idx))['weight_map']
        named = {os.path.join(folder, v) for v in wm.values()}
        if named != set(files):
            raise SystemExit(f'{folder}: index names {sorted(os.path.basename(f) for f in named)}, found {sorted(os.path.basename(f) for f in files)}')
    for fn in files:
        with safe_open(fn, framework='pt', device='cpu') as f:
            for k in f.keys():
                shapes[k] = tuple(f.get_slice(k).get_shape())
    return shapes


def report(title, missing, wrong, extra):
    ok = not missing and not wrong
    print(f'{title}: {"OK" if ok else "MISMATCH"}  ({len(missing)} missing, {len(wrong)} mis-shaped, {len(extra)} unexpected)')
    for k in missing[:20]:
        print('   missing   ', k)
    for k, got, want in wrong[:20]:
        print(f'   shape      {k}: {got}, expected {want}')
    for k in extra[:10]:
        print('   unexpected', k)
    return ok


def main(argv=None):
    from arcflow_amd.weights import check_state_shapes, expected_transformer_keys
    ap = argparse.ArgumentParser()
    ap.add_argument('snapshot')
    ap.add_argument('--family', choices=['flux', 'qwen'])
    ap.add_argument('--adapter', help='ArcFlow adapter directory (config.json + diffusion_pytorch_model.safetensors)')
    a = ap.parse_args(argv)
    tdir = os.path.join(a.snapshot, 'transformer')
    cfg = json.load(open(os.path.join(tdir, 'config.json')))
    family = a.family or ('qwen' if 'Qwen' in cfg.get('_class_name', '') else 'flux')
    ok = report(f'{tdir} [{family}]', *check_state_shapes(header_shapes(tdir), expected_transformer_keys(family, cfg)))
    if a.adapter:
        acfg = json.load(open(os.path.join(a.adapter, 'config.json')))
        K, L = acfg.get('num_gaussians', 16), acfg.get('logweights_channels', 4)
        exp = {k: v for k, v in expected_transformer_keys(family, cfg, student=True, K=K, L=L).items()
               if k.startswith(('proj_out_', 'norm_out.'))}
        shapes = {k[len('transformer.'):] if k.startswith('transformer.') else k: v for k, v in header_shapes(a.adapter).items()}
        lora = {k: v for k, v in shapes.items() if 'lora' in k}
        base = expected_transformer_keys(family, cfg)
        bad_lora = []
        for k, shp in lora.items():                # lora_A [r, in] / lora_B [out, r] of a linear that exists in the base model
            mod, ab = k.rsplit('.lora_', 1)
            w = base.get(mod + '.weight')
            if w is None or (ab.startswith('A') and shp[1] != w[1]) or (ab.startswith('B') and shp[0] != w[0]):
                bad_lora.append((k, shp, w))
        missing, wrong, extra = check_state_shapes({k: v for k, v in shapes.items() if 'lora' not in k}, exp)
        ok = report(f'{a.adapter} [heads + norm_out]', missing, wrong, extra) and ok
        print(f'   {len(lora)} LoRA tensors, {len(bad_lora)} not matching a base linear')
        for k, shp, w in bad_lora[:10]:
            print(f'   lora       {k}: {shp} vs base weight {w}')
        ok = ok and not bad_lora
    return 0 if ok else 1


if __name__ == '__main__':
    raise SystemExit(main())
