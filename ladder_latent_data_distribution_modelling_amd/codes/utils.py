"""Configuration and directory helpers.  Function names, printed messages, derived directory layout and exit codes follow the
reference's `codes/utils.py` (they are part of the drop-in surface: scripts and users read them); the implementation is this
project's own."""
import argparse
import json
import os
import sys
from datetime import datetime

# keys of the JSON that name an experiment's output folder, in the order the reference concatenates them (utils.py:52-60)
_RUN_NAME_KEYS = ("prior", "num_hidden_units", "code_size", "representation_size", "inner_activation", "n_layers_inner_VAE")


def get_config_from_json(json_file):
    with open(json_file) as fh:
        return json.load(fh)


def _run_name(cfg):
    return "prior-" + "-".join(str(cfg[k]) for k in _RUN_NAME_KEYS) + "-mixture-{}".format(cfg["n_mixtures"])


def _output_dirs(cfg, run):
    """load_dir == "default": ./experiments/<exp>/batch-<B>/<run>/{summary,result,checkpoint}/ ; otherwise results go under
    ./figures/<exp>/ and checkpoints are read from <load_dir>/<exp> (utils.py:65-76)."""
    exp = cfg["exp_name"]
    if cfg["load_dir"] == "default":
        root = os.path.join("./experiments/{}/batch-{}".format(exp, cfg["batch_size"]), run)
        return {k + "_dir": os.path.join(root, k + "/") for k in ("summary", "result", "checkpoint")}
    return dict(summary_dir="./figures/{}/summary/".format(exp), result_dir="./figures/{}/result/".format(exp),
                checkpoint_dir=os.path.join(cfg["load_dir"], exp))


def process_config(json_file):
    cfg = get_config_from_json(json_file)
    print("The current config is:\n{}\n".format(cfg))
    run = _run_name(cfg)
    print("Experiment results will be saved at:\n{}\n".format(run))
    cfg.update(_output_dirs(cfg, run))
    print("Models will be saved / loaded at:\n{}".format(cfg["checkpoint_dir"]))
    print("Results will be saved at:\n{}\n".format(cfg["result_dir"]))
    return cfg


def save_config(config):
    """<checkpoint_dir>training_config_<dd-Mon-YYYY-HH-MM>.txt holding the JSON dump."""
    target = "{}training_config_{:%d-%b-%Y-%H-%M}.txt".format(config["checkpoint_dir"], datetime.now())
    with open(target, "w") as fh:
        json.dump(config, fh)
    print("The current config is saved at {}".format(target))


def create_dirs(dirs):
    """Returns 0; a failure is reported and ends the process with status -1."""
    try:
        for path in dirs:
            os.makedirs(path, exist_ok=True)
    except Exception as err:  # noqa: BLE001
        print("Creating directories error: {0}".format(err))
        sys.exit(-1)
    return 0


# The reference counts `tf.trainable_variables(scope_name)` of the process-wide DEFAULT GRAPH (utils.py:96-113; called with the scope
# name only by base.py:438-444 and by the notebook, cell 19: `count_trainable_variables('interpolation')`).  The equivalent here: the
# most recently constructed model registers its parameter store as the default "graph", and other owners of trainable variables
# (codes/interpolation.py: the path points of the SLP optimisation) register their scope the same way.
_DEFAULT_GRAPH = {"model": None, "scopes": {}}


def set_default_model(model):
    """Called by BaseModel.__init__ (the analogue of building the model in TF's default graph)."""
    import weakref
    _DEFAULT_GRAPH["model"] = weakref.ref(model)


def register_trainable_scope(scope_name, n_parameters):
    """Trainable variables that live outside a model's parameter store (e.g. scope "interpolation")."""
    _DEFAULT_GRAPH["scopes"][str(scope_name)] = int(n_parameters)


def count_trainable_variables(scope_name, model=None):
    """Trainable parameter count under a variable scope ("encoder", "decoder", "sigma", "prior", "inner_sigma", "interpolation");
    same signature, printed line and return value as the reference (utils.py:96-113).  `model` (keyword, optional) selects a model other
    than the most recently constructed one.  Like tf.trainable_variables, the scope is a name PREFIX ("prior" also matches nothing
    else here because every scope is followed by "/")."""
    if not isinstance(scope_name, str):
        raise TypeError("count_trainable_variables(scope_name): scope_name must be a str (got %r)" % type(scope_name).__name__)
    m = model
    if m is None:
        ref = _DEFAULT_GRAPH["model"]
        m = ref() if ref is not None else None
    n = _DEFAULT_GRAPH["scopes"].get(scope_name, 0)
    if m is not None:
        n += m.engine.ps.num_params(scope_name + "/")
    print("The total number of trainable parameters in the {} model is: {}k.".format(scope_name, round(n / 1000, 2)))
    return n


def get_args():
    parser = argparse.ArgumentParser(description="LaDDer training on the MI355X HIP path")
    parser.add_argument("-c", "--config", metavar="C", default="None", help="The Configuration file")
    return parser.parse_args()
