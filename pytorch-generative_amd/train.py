"""Training entry point for the MI355X path (the reference's train.py:10-44, with its multi-GPU
launcher made to work).

    python train.py --model image_gpt --epochs 1 --batch-size 64 --logdir /tmp/run --gpus 1
    python train.py --model pixel_snail --gpus 8       # spawns one process per GPU (RCCL over xGMI)

`--gpus N > 1` starts N worker processes; worker r pins GPU r, joins a `nccl` (= RCCL) process
group on 127.0.0.1 and runs the model module's reproduce(..., n_gpus=N, device_id=r): every rank
draws its own batches (loader seeds are offset by the rank), gradients are summed with one flat RCCL
all-reduce per step (captured inside the step's hipGraph), rank 0 writes the checkpoints. (The reference's launcher hides all but one GPU from every worker and passes its
arguments in the wrong order, so `--gpus > 1` raises there — SURVEY.md §3.4; the parent here also
does not fall through into a second, single-GPU run.)
"""

import argparse
import os

import torch

from pytorch_generative_amd.models import autoregressive, vae

MODEL_DICT = {
    "beta_vae": vae.beta_vae,
    "gated_pixel_cnn": autoregressive.gated_pixel_cnn,
    "image_gpt": autoregressive.image_gpt,
    "pixel_cnn": autoregressive.pixel_cnn,
    "pixel_cnn_pp": autoregressive.pixel_cnn_pp,
    "pixel_snail": autoregressive.pixel_snail,
    "vae": vae.vae,
    "vd_vae": vae.vd_vae,
    "vq_vae": vae.vq_vae,
    "vq_vae_2": vae.vq_vae_2,
}


def _worker(rank, model, epochs, batch_size, logdir, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # development hooks (1-GPU box): PG_FORCE_DEVICE pins every rank to one GPU, PG_DIST_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device); PG_TRAIN_DUMP=<dir> makes every rank save
    # its final parameters there (tests/test_gpu_dp.py compares the ranks)
    device = int(os.environ.get("PG_FORCE_DEVICE", rank))
    backend = os.environ.get("PG_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
    torch.cuda.set_device(device)
    if backend == "nccl":
        torch.distributed.init_process_group(backend="nccl", world_size=world, rank=rank,
                                             device_id=torch.device("cuda", device))
    else:
        torch.distributed.init_process_group(backend=backend, world_size=world, rank=rank)
    try:
        t = MODEL_DICT[model].reproduce(epochs, batch_size, logdir, n_gpus=world, device_id=device)
        dump = os.environ.get("PG_TRAIN_DUMP")
        if dump:
            torch.save({"params": {k: v.detach().cpu() for k, v in t.model.named_parameters()},
                        "step": t._step, "first_batch": next(iter(t.train_loader))[0].cpu()},
                       os.path.join(dump, f"rank{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


def main(args):
    if args.gpus > 1:
        torch.multiprocessing.spawn(
            _worker, (args.model, args.epochs, args.batch_size, args.logdir, args.gpus, args.port),
            nprocs=args.gpus)
        return
    MODEL_DICT[args.model].reproduce(args.epochs, args.batch_size, args.logdir, n_gpus=1, device_id=0)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--model", type=str, default="image_gpt", choices=sorted(MODEL_DICT),
                        help="the model to train")
    parser.add_argument("--epochs", type=int, default=1, help="number of training epochs")
    parser.add_argument("--batch-size", type=int, default=128,
                        help="the per-GPU training and evaluation batch size")
    parser.add_argument("--logdir", type=str, default="/tmp/run",
                        help="directory for checkpoints and TensorBoard summaries")
    parser.add_argument("--gpus", type=int, default=1, help="number of MI355X GPUs (one process each)")
    parser.add_argument("--port", type=int, default=29533, help="rendezvous port on 127.0.0.1")
    main(parser.parse_args())
