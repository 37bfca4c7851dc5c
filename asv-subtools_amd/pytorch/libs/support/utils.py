# -*- coding:utf-8 -*-
"""Support helpers of the extraction path (the subset of the reference's
/root/reference/pytorch/libs/support/utils.py that `pipeline/onestep/extract_embeddings.py`
and the model blueprints use): blueprint loading (163-186), nnet.config (189-202), device
selection (56-102), `to_device` / `for_device_free` (105-160) and `assign_params_dict`
(319-356).  Training-side helpers (DDP/Horovod init, optimiser plumbing) are out of scope.
"""

import copy
import csv
import logging
import os
import sys

import numpy as np
import torch

logger = logging.getLogger("libs")
logger.setLevel(logging.INFO)
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("%(asctime)s [%(pathname)s:%(lineno)s] %(message)s"))
    logger.addHandler(_h)


def to_bool(variable):
    """'true'/'false' strings (any case) or bools -> bool."""
    if isinstance(variable, bool):
        return variable
    if isinstance(variable, str):
        low = variable.strip().lower()
        if low in ("true", "false"):
            return low == "true"
    raise ValueError("Expected bool or 'true'/'false' string, but got {}.".format(variable))


def parse_gpu_id_option(gpu_id):
    """'0', '0,1', '0-1-2', 0, [0, 1] -> list of ints."""
    if isinstance(gpu_id, (list, tuple)):
        return [int(x) for x in gpu_id]
    if isinstance(gpu_id, int):
        return [gpu_id]
    if isinstance(gpu_id, str):
        text = gpu_id.replace("-", " ").replace(",", " ")
        return [int(x) for x in text.split()]
    raise TypeError("Expected str, int or list/tuple, bug got {}.".format(gpu_id))


def use_ddp():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def auto_select_gpu():
    """The reference shells out to nvidia-smi (GPU_Manager.py:39,73), which does not exist on
    ROCm.  One process per GPU is the deployment model here: honour LOCAL_RANK when a
    launcher set it, else take the device with the most free memory."""
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"])
    best, best_free = 0, -1
    for i in range(torch.cuda.device_count()):
        try:
            free, _ = torch.cuda.mem_get_info(i)
        except Exception:
            free = 0
        if free > best_free:
            best, best_free = i, free
    return best


def select_model_device(model, use_gpu, gpu_id="", benchmark=False):
    """Moves `model` to CPU, or to the selected ROCm device (reference utils.py:56-102).
    DDP / Horovod wrapping is a training concern and is not done here."""
    model.cpu()
    if to_bool(use_gpu):
        if not torch.cuda.is_available():
            raise RuntimeError("use_gpu is true but no ROCm device is visible to torch")
        if gpu_id == "" or gpu_id is None:
            logger.info("The use_gpu is true and gpu id is not specified, so select gpu device automatically.")
            ids = [auto_select_gpu()]
        else:
            ids = parse_gpu_id_option(gpu_id)
        dev = ids[torch.distributed.get_rank() % len(ids)] if use_ddp() else ids[0]
        torch.cuda.set_device(dev)
        model.cuda()
    return model


def to_device(device_object, tensor):
    if isinstance(device_object, torch.nn.Module):
        device = next(device_object.parameters()).device
    elif isinstance(device_object, torch.Tensor):
        device = device_object.device
    else:
        raise TypeError("Expected a module or tensor, got {}".format(type(device_object)))
    return tensor.to(device)


def get_device(model):
    assert isinstance(model, torch.nn.Module)
    return next(model.parameters()).device


def get_tensors(tensor_sets):
    """Flattens nested lists/tuples of tensors / ndarrays into a list of tensors."""
    out = []
    for obj in tensor_sets:
        if isinstance(obj, torch.Tensor):
            out.append(obj)
        elif isinstance(obj, np.ndarray):
            out.append(torch.from_numpy(obj))
        elif isinstance(obj, (list, tuple)):
            out.extend(get_tensors(obj))
        else:
            out.append(obj)          # symbolic handles of the HIP recorder pass through untouched
    return out


def for_device_free(function):
    """Decorator: move tensor arguments to the module's device (reference utils.py:149-160)."""
    def wrapper(self, *tensor_sets):
        moved = [to_device(self, t) if isinstance(t, torch.Tensor) else t for t in get_tensors(tensor_sets)]
        return function(self, *moved)
    return wrapper


def create_model_from_py(model_blueprint, model_creation=""):
    """Imports the blueprint file and evaluates the creation string, e.g.
    create_model_from_py('model/xvector.py', 'Xvector(30, 10, training=False)')."""
    if not os.path.exists(model_blueprint):
        raise TypeError("Expected {} to exist.".format(model_blueprint))
    if os.path.getsize(model_blueprint) == 0:
        raise TypeError("There is nothing in {}.".format(model_blueprint))
    sys.path.insert(0, os.path.dirname(model_blueprint))
    module_name = os.path.basename(model_blueprint).split(".")[0]
    model_module = __import__(module_name)
    if model_creation == "":
        return model_module
    return eval("model_module.{0}".format(model_creation))


def write_nnet_config(model_blueprint, model_creation, nnet_config):
    """Two ';'-separated rows: model_blueprint;<path> and model_creation;<ctor string>."""
    with open(nnet_config, "w", newline="") as f:
        w = csv.writer(f, delimiter=";", lineterminator="\n")
        w.writerow(["model_blueprint", model_blueprint])
        w.writerow(["model_creation", model_creation])
    logger.info("Save nnet_config to {0} done.".format(nnet_config))


def read_nnet_config(nnet_config):
    logger.info("Read nnet_config from {0}".format(nnet_config))
    rows = {}
    with open(nnet_config, "r", newline="") as f:
        for row in csv.reader(f, delimiter=";"):
            if len(row) >= 2:
                rows[row[0]] = row[1]
    return rows["model_blueprint"], rows["model_creation"]


def assign_params_dict(default_params, params, force_check=False, support_unknow=False):
    """default <= params for matching keys (recursively for dict values); keys unknown to
    `default_params` are dropped unless support_unknow (reference utils.py:319-356 - this is
    why e.g. `**ecapa_params` can be splatted into every layer)."""
    merged = copy.deepcopy(default_params)
    if force_check:
        for key in params:
            if key not in merged:
                raise ValueError("The params key {0} is not in default params".format(key))
    for k, v in list(merged.items()):
        if k not in params:
            continue
        new = params[k]
        if isinstance(v, type(new)):
            merged[k] = assign_params_dict(v, new, force_check, support_unknow) if isinstance(v, dict) else new
        elif isinstance(v, float) and isinstance(new, int):
            merged[k] = new * 1.0
        elif v is None or new is None:
            merged[k] = new
        else:
            raise ValueError("The value type of default params [{0}] is not equal to [{1}] of params for k={2}".format(
                type(v), type(new), k))
    if support_unknow and not force_check:
        for k, v in params.items():
            if k not in merged:
                merged[k] = v
    return merged


def iterator_to_params_str(iterator, sep=","):
    return sep.join("'{}'".format(x) if isinstance(x, str) else str(x) for x in iterator)


def dict_to_params_str(dict, auto=True, connect="=", sep=","):
    parts = []
    for k, v in dict.items():
        if auto and isinstance(v, str):
            v = "'{}'".format(v)
        parts.append("{}{}{}".format(k, connect, v))
    return sep.join(parts)


def key_to_value(adict, key, return_none=True):
    assert isinstance(adict, dict)
    if key in adict:
        return adict[key]
    return None if return_none else key


def set_all_seed(seed=None, deterministic=True):
    if seed is not None:
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
