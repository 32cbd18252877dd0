#!/usr/bin/env python3
"""Counterpart of the reference harness scripts/sample.py:17-57 on the HIP path.

Same flow: seed, load pipeline, `.to(device)`, loop over conditions [0, 1, None] calling
`pipeline.sample(16, (8,32,32), guidance_scale=8, condition=..., un_cond=None, steps=150, use_ddim=True)`,
then `(x+1)/2`, clamp, `save_image(normalize=True, scale_each=True)` and the |class1 - class0| difference image.
Only the import line and the checkpoint source differ; `--synthetic` builds the published architecture with
seeded synthetic weights because no checkpoint/dataset exists offline.
"""
import argparse
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import DiffusionPipeline


def rgb2gray(img):
    return ((0.3 * img[:, 0]) + (0.59 * img[:, 1]) + (0.11 * img[:, 2]))[:, None]


def normalize(img):
    return torch.stack([(b - b.min()) / (b.max() - b.min()) for b in img])


def save_image(tensor, path, nrow=8, normalize=False, scale_each=False, padding=2):
    """torchvision.utils.save_image semantics used by the reference: per-image min-max (scale_each), grid, PNG."""
    t = tensor.detach().float().cpu()
    if t.shape[1] == 1:
        t = t.expand(-1, 3, -1, -1)
    if normalize:
        if scale_each:
            t = torch.stack([(b - b.min()) / (b.max() - b.min()).clamp_min(1e-5) for b in t])
        else:
            t = (t - t.min()) / (t.max() - t.min()).clamp_min(1e-5)
    n, c, h, w = t.shape
    xm = min(nrow, n)
    ym = int(math.ceil(n / xm))
    grid = torch.zeros((c, ym * (h + padding) + padding, xm * (w + padding) + padding))
    for k in range(n):
        y, x = divmod(k, xm)
        grid[:, y * (h + padding) + padding:y * (h + padding) + padding + h, x * (w + padding) + padding:x * (w + padding) + padding + w] = t[k]
    arr = (grid.clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).numpy()
    try:
        from PIL import Image
        Image.fromarray(arr).save(str(path))
    except ImportError:  # no PIL in the image: keep the pixels anyway
        import numpy as np
        np.save(str(path) + ".npy", arr)


def synthetic_pipeline():
    """the published architecture with seeded weights (no checkpoint exists offline): medfusion_amd/published.py"""
    from medfusion_amd.published import build_published_pipeline
    return build_published_pipeline(None, num_classes=2)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default="runs/2022_12_12_171357_chest_diffusion/last.ckpt")
    ap.add_argument("--latent-embedder-ckpt", default=None, help="VAE checkpoint when the path baked into --ckpt does not exist here")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--out", default="results/CheXpert/samples")
    args = ap.parse_args()
    path_out = Path.cwd() / args.out
    path_out.mkdir(parents=True, exist_ok=True)

    torch.manual_seed(0)
    device = torch.device("cuda")
    ckpt_kw = {"latent_embedder_checkpoint": args.latent_embedder_ckpt} if args.latent_embedder_ckpt else {}
    pipeline = synthetic_pipeline() if args.synthetic else DiffusionPipeline.load_from_checkpoint(args.ckpt, **ckpt_kw)
    pipeline.to(device)

    steps, use_ddim, images, n_samples = args.steps, True, {}, args.n
    raw = {}
    for cond in [0, 1, None]:
        torch.manual_seed(0)
        condition = torch.tensor([cond] * n_samples, device=device) if cond is not None else None
        un_cond = None
        results = pipeline.sample(n_samples, (8, 32, 32), guidance_scale=8, condition=condition, un_cond=un_cond, steps=steps, use_ddim=use_ddim)
        raw[str(cond)] = results.cpu()
        results = (results + 1) / 2
        results = results.clamp(0, 1)
        save_image(results, path_out / f"test_{cond}.png", nrow=int(math.sqrt(results.shape[0])), normalize=True, scale_each=True)
        images[cond] = results
    diff = torch.abs(normalize(rgb2gray(images[1])) - normalize(rgb2gray(images[0])))
    save_image(diff, path_out / "diff.png", nrow=int(math.sqrt(results.shape[0])), normalize=True, scale_each=True)
    torch.save(raw, path_out / "samples_raw.pt")   # the tensors behind the PNGs (what the harness test pins)
    print(f"wrote {path_out}")
