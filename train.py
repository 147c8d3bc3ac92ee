"""python3 train.py --config codes/<exp>_config.json   -- same CLI as the reference's train.py:18-74.

Single GPU:   python3 train.py -c codes/celeba_config.json
Data parallel (one process per GPU, RCCL):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py -c codes/celeba_config.json
"""
import os

from codes.data_loader import DataGenerator
from codes.models import MNISTModel_digit, MNISTModel_fashion, CelebAModel_densenet
from codes.trainers import MNISTTrainer_joint_training, CelebATrainer_joint_training
from codes.session import Session          # stands in for tf.Session (train.py:41-47 of the reference)
from codes.utils import process_config, create_dirs, get_args, save_config


def main():
    try:
        args = get_args()
        config = process_config(args.config)
    except Exception:  # noqa: BLE001 - reference behaviour (train.py:21-27)
        print("missing or invalid arguments")
        exit(0)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    create_dirs([config["result_dir"], config["checkpoint_dir"]])
    save_config(config)
    sess = Session()
    data = DataGenerator(config, sess)
    model = {"mnist_digit": MNISTModel_digit, "mnist_fashion": MNISTModel_fashion, "celeba": CelebAModel_densenet}[config["exp_name"]](config)
    print("Created a VAE model.")
    print("The current dataset is {}, num hidden units: {}.\n".format(config["exp_name"], config["num_hidden_units"]))
    if config["TRAIN_VAE"] or config["TRAIN_sigma"] or config["TRAIN_prior"]:
        if config["exp_name"] in ("mnist_digit", "mnist_fashion"):
            trainer = MNISTTrainer_joint_training(sess, model, data, config)
        else:
            trainer = CelebATrainer_joint_training(sess, model, data, config)
        model.load(sess, model="VAE")
        if config["prior"] in ("ours", "hierarchical", "vampPrior"):
            model.load(sess, model="prior")
        if config["num_epochs"] > 0:
            trainer.train()


if __name__ == "__main__":
    main()
