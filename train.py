"""Command-line entry of the LaDDer HIP path.  The CLI is the reference's: `python3 train.py --config codes/<exp>_config.json`
(same flag, same JSON keys, same messages and exit codes as the reference script it replaces).

    1 GPU:        python3 train.py -c codes/celeba_config.json
    8 GPUs (DP):  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py -c codes/celeba_config.json
"""
import os
import sys

import codes.models as models
import codes.trainers as trainers
from codes import utils
from codes.data_loader import DataGenerator
from codes.session import Session

# exp_name -> (model class, trainer class)
REGISTRY = {
    "mnist_digit": (models.MNISTModel_digit, trainers.MNISTTrainer_joint_training),
    "mnist_fashion": (models.MNISTModel_fashion, trainers.MNISTTrainer_joint_training),
    "celeba": (models.CelebAModel_densenet, trainers.CelebATrainer_joint_training),
}
PRIORS_WITH_OWN_CHECKPOINT = ("ours", "hierarchical", "vampPrior")


def read_config():
    """Any failure while parsing the arguments or the JSON prints one line and exits with status 0 (the reference's behaviour)."""
    try:
        return utils.process_config(utils.get_args().config)
    except Exception:  # noqa: BLE001
        print("missing or invalid arguments")
        sys.exit(0)


def join_process_group():
    """One process per GPU under torch.distributed.run; a plain `python3 train.py` stays single-process."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))     # (bound to its GPU at once: no lazy communicator set-up inside the first collective)


def wants_training(cfg):
    return bool(cfg["TRAIN_VAE"] or cfg["TRAIN_sigma"] or cfg["TRAIN_prior"])


def main():
    cfg = read_config()
    join_process_group()
    utils.create_dirs([cfg["result_dir"], cfg["checkpoint_dir"]])
    utils.save_config(cfg)

    session = Session()                      # takes the place of tf.Session; trainers accept and ignore it
    model_cls, trainer_cls = REGISTRY[cfg["exp_name"]]
    data = DataGenerator(cfg, session)
    model = model_cls(cfg)
    print("Created a VAE model.")
    print("The current dataset is {}, num hidden units: {}.\n".format(cfg["exp_name"], cfg["num_hidden_units"]))
    if not wants_training(cfg):
        return

    trainer = trainer_cls(session, model, data, cfg)
    for which in ("VAE",) + (("prior",) if cfg["prior"] in PRIORS_WITH_OWN_CHECKPOINT else ()):
        model.load(session, model=which)     # resumes from <checkpoint_dir>/{vae,prior}-model if present
    if cfg["num_epochs"] > 0:
        trainer.train()


if __name__ == "__main__":
    main()
