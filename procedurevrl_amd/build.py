"""Model registry + `build_model(cfg)`.

Same surface as the reference's `lib/models/build.py:8,17-54` (an fvcore `Registry` named MODEL whose
entries are called as `obj(cfg)` and return an `nn.Module`).  Data parallelism differs by design: the
reference wraps the module in `DistributedDataParallel(find_unused_parameters=True)` (build.py:49-53);
here gradients live in one flat fp32 buffer (engine.GradStore) that `distributed.GradReducer`
all-reduces over RCCL in large chunks, so `build_model` returns the bare module for any NUM_GPUS.
"""
import torch


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(fn_or_cls):
                self._do_register(fn_or_cls.__name__, fn_or_cls)
                return fn_or_cls
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map


MODEL_REGISTRY = Registry("MODEL")


def build_model(cfg, gpu_id=None):
    import os
    if torch.cuda.is_available():
        # (PVRL_SINGLE_DEVICE: functional tests run the N-process path with every rank on one GPU, gloo backend)
        assert cfg.NUM_GPUS <= torch.cuda.device_count() or os.environ.get("PVRL_SINGLE_DEVICE"), \
            "Cannot use more GPU devices than available"
    else:
        assert cfg.NUM_GPUS == 0, "Cuda is not available. Please set `NUM_GPUS: 0 for running on CPUs."
    from . import vit, mvit  # noqa: F401  (register the models)
    model = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    if cfg.NUM_GPUS:
        cur_device = torch.cuda.current_device() if gpu_id is None else gpu_id
        model = model.cuda(device=cur_device)
    return model
