"""Config / directory helpers with the reference's names and behaviour (codes/utils.py:11-124)."""
import argparse
import json
import os
from datetime import datetime


def get_config_from_json(json_file):
    """JSON file -> dict (codes/utils.py:11-21)."""
    with open(json_file, "r") as f:
        return json.load(f)


def save_config(config):
    """Dump the config next to the checkpoints as training_config_<timestamp>.txt (codes/utils.py:24-37)."""
    stamp = datetime.now().strftime("%d-%b-%Y-%H-%M")
    filename = config["checkpoint_dir"] + "training_config_{}.txt".format(stamp)
    with open(filename, "w") as f:
        f.write(json.dumps(config))
    print("The current config is saved at {}".format(filename))


def process_config(json_file):
    """Read the JSON and derive summary_dir / result_dir / checkpoint_dir exactly as codes/utils.py:40-77."""
    config = get_config_from_json(json_file)
    print("The current config is:\n{}\n".format(config))
    save_name = "prior-{}-{}-{}-{}-{}-{}-mixture-{}".format(
        config["prior"], config["num_hidden_units"], config["code_size"], config["representation_size"],
        config["inner_activation"], config["n_layers_inner_VAE"], config["n_mixtures"])
    print("Experiment results will be saved at:\n{}\n".format(save_name))
    if config["load_dir"] == "default":
        save_dir = "./experiments/{}/batch-{}".format(config["exp_name"], config["batch_size"])
        config["summary_dir"] = os.path.join(save_dir, save_name, "summary/")
        config["result_dir"] = os.path.join(save_dir, save_name, "result/")
        config["checkpoint_dir"] = os.path.join(save_dir, save_name, "checkpoint/")
    else:
        save_dir = config["load_dir"]
        config["summary_dir"] = "./figures/{}/summary/".format(config["exp_name"])
        config["result_dir"] = "./figures/{}/result/".format(config["exp_name"])
        config["checkpoint_dir"] = os.path.join(save_dir, config["exp_name"])
    print("Models will be saved / loaded at:\n{}".format(config["checkpoint_dir"]))
    print("Results will be saved at:\n{}\n".format(config["result_dir"]))
    return config


def create_dirs(dirs):
    """Create the directories; exit(-1) on failure like codes/utils.py:80-93."""
    try:
        for d in dirs:
            if not os.path.exists(d):
                os.makedirs(d)
        return 0
    except Exception as err:  # noqa: BLE001 - mirrors the reference's behaviour
        print("Creating directories error: {0}".format(err))
        exit(-1)


def count_trainable_variables(model, scope_name):
    """Number of trainable parameters under a variable scope (codes/utils.py:96-113)."""
    total = model.engine.ps.num_params(scope_name + "/")
    print("The total number of trainable parameters in the {} model is: {}k.".format(scope_name, round(total / 1000, 2)))
    return total


def get_args():
    """`-c/--config` (codes/utils.py:116-124)."""
    p = argparse.ArgumentParser(description=__doc__)
    p.add_argument("-c", "--config", metavar="C", default="None", help="The Configuration file")
    return p.parse_args()
