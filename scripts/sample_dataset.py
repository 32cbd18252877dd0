#!/usr/bin/env python3
"""Counterpart of the reference's bulk generator scripts/helpers/sample_dataset.py:16-57 on the HIP path.

Same flow: load the pipeline, `.to(device)`; for every `steps` setting and every (name, label): `torch.manual_seed(0)`, then
`n_samples` images in chunks of `sample_batch` with `pipeline.sample(len(chunk), (8, 32, 32), guidance_scale=1, condition=label,
un_cond=1 - label, steps=steps)`, each image clipped to [-1, 1], scaled to uint8 and saved as `fake_<counter>.png`.
Differences, outside the arithmetic: the clip / uint8 / channel-last conversion runs on the device and the pixels leave through pinned
buffers with an asynchronous copy while the next chunk samples (medfusion_amd/egress.py); `--synthetic` builds the published
architecture with seeded weights because no checkpoint exists offline; the run prints one JSON line with the images/s it reached
with and without the egress.
"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from medfusion_amd import DiffusionPipeline
from medfusion_amd.egress import AsyncImageWriter, save_png


def chunks(lst, n):
    for i in range(0, len(lst), n):
        yield lst[i:i + n]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default="runs/2022_12_12_171357_chest_diffusion/last.ckpt")
    ap.add_argument("--latent-embedder-ckpt", default=None, help="VAE checkpoint when the path baked into --ckpt does not exist here")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--steps-list", default="50,100,150,200,250")
    ap.add_argument("--labels", default="No_Cardiomegaly:0,Cardiomegaly:1")
    ap.add_argument("--n-samples", type=int, default=7869)
    ap.add_argument("--sample-batch", type=int, default=200)
    ap.add_argument("--out", default="generated_diffusion3")
    ap.add_argument("--no-files", action="store_true", help="run the egress (device conversion + async copy) but skip the PNG encoder")
    ap.add_argument("--compare-no-egress", action="store_true", help="also time the same generation with the images left on the device")
    ap.add_argument("--writer-threads", type=int, default=2, help="PNG encoder threads of the asynchronous writer")
    args = ap.parse_args()

    device = torch.device("cuda")
    if args.synthetic:
        from medfusion_amd.published import build_published_pipeline
        pipeline = build_published_pipeline(None, num_classes=2)
    else:
        kw = {"latent_embedder_checkpoint": args.latent_embedder_ckpt} if args.latent_embedder_ckpt else {}
        pipeline = DiffusionPipeline.load_from_checkpoint(args.ckpt, **kw)
    pipeline.to(device)
    labels = [(kv.split(":")[0], int(kv.split(":")[1])) for kv in args.labels.split(",")]
    cfg = 1
    stats = {}
    for egress in ([True, False] if args.compare_no_egress else [True]):
        total, t_all = 0, 0.0
        for steps in [int(v) for v in args.steps_list.split(",")]:
            for name, label in labels:
                path_out = Path(f"{args.out}_{steps}") / name
                if egress and not args.no_files:
                    path_out.mkdir(parents=True, exist_ok=True)
                writer = AsyncImageWriter(device, sink=None if args.no_files else save_png, threads=args.writer_threads) if egress else None
                torch.manual_seed(0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                counter = 0
                for chunk in chunks(list(range(args.n_samples)), args.sample_batch):
                    condition = torch.tensor([label] * len(chunk), device=device)
                    un_cond = torch.tensor([1 - label] * len(chunk), device=device)
                    results = pipeline.sample(len(chunk), (8, 32, 32), guidance_scale=cfg, condition=condition, un_cond=un_cond, steps=steps)
                    if writer is not None:
                        writer.submit(results, [path_out / f"fake_{counter + i}.png" for i in range(len(chunk))])
                    counter += len(chunk)
                if writer is not None:
                    writer.close()
                torch.cuda.synchronize()
                t_all += time.perf_counter() - t0
                total += counter
        stats["with_egress" if egress else "images_left_on_device"] = {"images": total, "seconds": round(t_all, 3), "images_per_s": round(total / t_all, 3)}
    out = {"bulk_generation": stats, "sample_batch": args.sample_batch, "guidance_scale": cfg, "n_samples_per_class": args.n_samples, "steps": args.steps_list,
           "png_files_written": not args.no_files, "writer_threads": args.writer_threads}
    if "with_egress" in stats and "images_left_on_device" in stats:
        out["egress_overlap"] = round(stats["with_egress"]["images_per_s"] / stats["images_left_on_device"]["images_per_s"], 4)   # 1.0 = the egress is free
    print(json.dumps(out))
