"""yacs-compatible config node + the default key tree of the hot path.

Mirrors the surface of the reference's `lib/config/defaults.py` (`get_cfg()` :1073-1077 returning an
fvcore/yacs `CfgNode`) closely enough that the reference's eight yaml files under `configs/` load
unchanged, including yacs' `literal_eval` decoding of string values such as `'1e-4'` and
`"(2, 4, 4)"` (configs/HowTo100M/procedurevrl_mvitv2_adamw.yaml:37-39), type-checked merges,
`merge_from_list(["KEY.SUB", "VAL", ...])` CLI overrides (lib/utils/parser.py:80) and `dump()`
(stored in checkpoints, lib/utils/checkpoint.py:129).  fvcore / yacs are not available in the
target image, hence this self-contained implementation.
"""
import ast
import copy

import yaml

_VALID = (tuple, list, str, int, float, bool, type(None))


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute access ---------------------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode.IMMUTABLE]:
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def freeze(self):
        self._immutable(True)

    def defrost(self):
        self._immutable(False)

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._immutable(flag)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        new = CfgNode()
        for k, v in self.items():
            new[k] = copy.deepcopy(v, memo)
        new.__dict__[CfgNode.IMMUTABLE] = self.__dict__[CfgNode.IMMUTABLE]
        return new

    # (de)serialisation ----------------------------------------------------------------
    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kwargs):
        def plain(v):
            if isinstance(v, CfgNode):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, tuple):
                return [plain(x) for x in v]
            if isinstance(v, list):
                return [plain(x) for x in v]
            if not isinstance(v, _VALID):       # e.g. an in-memory synthetic label-embedding tensor
                shape = tuple(getattr(v, "shape", ()))
                return f"<{type(v).__name__} {shape}>"
            return v
        return yaml.safe_dump(plain(self), **kwargs)

    @staticmethod
    def load_cfg(text):
        return CfgNode(yaml.safe_load(text) or {})

    # merging --------------------------------------------------------------------------
    @staticmethod
    def _decode(v):
        """yacs `_decode_cfg_value`: dict -> CfgNode, str -> literal_eval when it parses."""
        if isinstance(v, dict):
            return CfgNode(v)
        if not isinstance(v, str):
            return v
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v

    @staticmethod
    def _coerce(replacement, original, key, full_key):
        """yacs `_check_and_coerce_cfg_value_type`."""
        ot, rt = type(original), type(replacement)
        if rt == ot or original is None or replacement is None:
            return replacement
        for from_t, to_t in ((list, tuple), (tuple, list)):
            if rt == from_t and ot == to_t:
                return to_t(replacement)
        if ot is float and rt is int:  # fvcore/yacs allow int -> float
            return float(replacement)
        raise ValueError(f"Type mismatch ({ot} vs. {rt}) with values ({original} vs. {replacement}) for config key: {full_key}")

    def _merge(self, other, stack):
        for k, v_ in other.items():
            full = ".".join(stack + [k])
            v = CfgNode._decode(copy.deepcopy(v_))
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, CfgNode):
                    raise ValueError(f"Expected a mapping for config key: {full}")
                self[k]._merge(v, stack + [k])
            else:
                self[k] = CfgNode._coerce(v, self[k], k, full)

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_file(self, path):
        with open(path, "r") as f:
            loaded = yaml.safe_load(f) or {}
        base = loaded.pop("_BASE_", None)
        if base:
            import os
            self.merge_from_file(os.path.join(os.path.dirname(path), base))
        self._merge(CfgNode(loaded), [])

    def merge_from_list(self, cfg_list):
        if len(cfg_list) % 2 != 0:
            raise ValueError(f"Override list has odd length: {cfg_list}; it must be a list of pairs")
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            parts = full_key.split(".")
            d = self
            for sub in parts[:-1]:
                if sub not in d:
                    raise KeyError(f"Non-existent key: {full_key}")
                d = d[sub]
            if parts[-1] not in d:
                raise KeyError(f"Non-existent key: {full_key}")
            d[parts[-1]] = CfgNode._coerce(CfgNode._decode(v), d[parts[-1]], parts[-1], full_key)


# Default tree.  Only the sections that the shipped yaml files or the hot path read are kept
# (the reference's ResNet/X3D/AVA/Detection/Demo/Tensorboard sections configure code that is out of
# scope here); values equal the reference defaults for the keys kept.
_DEFAULTS = {
    "TASK": "Classification",
    "BN": {"USE_PRECISE_STATS": False, "NUM_BATCHES_PRECISE": 200, "WEIGHT_DECAY": 0.0, "NORM_TYPE": "batchnorm",
           "NUM_SPLITS": 1, "NUM_SYNC_DEVICES": 1},
    "DEV": {"ENABLE": False, "LOAD_DUMMY_DATA": False, "CLIP_LINKING": False, "CLIP_VIS_FEAT_PATH": "",
            "CLIP_VIS_FEAT_INPUT": False, "MATCH_LANG_EMB": False, "TEST_LANG_EMB": "", "TEMP": 0.02,
            "ZERO_SHOT_ENABLED": False, "ORDER_PRETRAIN_ENABLED": False, "ORDER_PRETRAIN_MAX_LEN": 9,
            "ORDER_FIX_RECOGNITION": False, "ORDER_STRIDE": 2, "ORDER_TFM_LAYERS": 4, "ORDER_RECOG_BATCH": 9,
            "INPUT_NEXT_CLIP": False, "EDIT_DISTANCE": 0, "EPIC_USE_FRAME_LOADER": False},
    "TRAIN": {"ENABLE": True, "DATASET": "kinetics", "LABEL_EMB": "", "FINETUNE": False, "SEP_LR": False,
              "LINEAR": False, "EVAL": False, "MULT": 1.0, "TEXT": "", "TEXT_SAMPLE": 0, "EPOCH_MUL": 1,
              "TEXT_EMB": "", "TOPK": 5, "BATCH_SIZE": 64, "EVAL_PERIOD": 10, "CHECKPOINT_PERIOD": 10,
              "AUTO_RESUME": True, "CHECKPOINT_FILE_PATH": "", "CHECKPOINT_TYPE": "pytorch",
              "CHECKPOINT_INFLATE": False, "CHECKPOINT_EPOCH_RESET": False, "CHECKPOINT_CLEAR_NAME_PATTERN": ()},
    "TEST": {"ENABLE": True, "DATASET": "kinetics", "BATCH_SIZE": 8, "CHECKPOINT_FILE_PATH": "",
             "NUM_ENSEMBLE_VIEWS": 10, "NUM_SPATIAL_CROPS": 3, "CHECKPOINT_TYPE": "pytorch", "SAVE_RESULTS_PATH": "",
             "SAVE_PREDICT_PATH": "", "SPLIT": ""},
    "MVIT": {"MODE": "conv", "POOL_FIRST": False, "CLS_EMBED_ON": True, "PATCH_KERNEL": [3, 7, 7],
             "PATCH_STRIDE": [2, 4, 4], "PATCH_PADDING": [2, 4, 4], "PATCH_2D": False, "EMBED_DIM": 96, "NUM_HEADS": 1,
             "MLP_RATIO": 4.0, "QKV_BIAS": True, "DROPPATH_RATE": 0.1, "LAYER_SCALE_INIT_VALUE": 0.0, "DEPTH": 16,
             "NORM": "layernorm", "DIM_MUL": [], "HEAD_MUL": [], "POOL_KV_STRIDE": [], "POOL_KV_STRIDE_ADAPTIVE": None,
             "POOL_Q_STRIDE": [], "POOL_KVQ_KERNEL": None, "ZERO_DECAY_POS_CLS": True, "NORM_STEM": False,
             "SEP_POS_EMBED": False, "DROPOUT_RATE": 0.0, "USE_ABS_POS": True, "REL_POS_SPATIAL": False,
             "REL_POS_TEMPORAL": False, "REL_POS_ZERO_INIT": False, "RESIDUAL_POOLING": False, "DIM_MUL_IN_ATT": False,
             "SEPARATE_QKV": False, "HEAD_INIT_SCALE": 1.0, "USE_MEAN_POOLING": False, "USE_FIXED_SINCOS_POS": False},
    "MODEL": {"ARCH": "slowfast", "MODEL_NAME": "SlowFast", "NUM_CLASSES": 400, "LOSS_FUNC": "cross_entropy",
              "SINGLE_PATHWAY_ARCH": ["c2d", "i3d", "slow", "x3d", "vit", "swin3d", "mvit"],
              "MULTI_PATHWAY_ARCH": ["slowfast"], "DROPOUT_RATE": 0.5, "DROPCONNECT_RATE": 0.0, "FC_INIT_STD": 0.01,
              "HEAD_ACT": "softmax", "ACT_CHECKPOINT": False, "PRETRAINED": True, "MLP": 0, "TEXT_MODEL": "",
              "TEXT_LP": False, "MAX_LEN": 64, "MIN_LEN": 0, "VIDEO_ONLY": False, "NUM_SEG": 0, "EXTRA_TR": "",
              "DROP_E": 0.0, "EXTRA_POS": False, "RET_HEAD": 0, "PRE_CLASSES": 0, "HEAD_T": True, "RET_POS": False,
              "RET_POS_MUL": False, "DROP_PATH": 0.1},
    "TIMESFORMER": {"ATTENTION_TYPE": "divided_space_time", "PRETRAINED_MODEL": "", "DEPTH": 12},
    "MIXUP": {"ENABLED": False, "ALPHA": 0.8, "CUTMIX_ALPHA": 1.0, "CUTMIX_MINMAX": None, "PROB": 1.0,
              "SWITCH_PROB": 0.5, "MODE": "batch"},
    "EMA": {"ENABLED": False},
    "DATA": {"PATH_TO_DATA_DIR": "", "PATH_LABEL_SEPARATOR": " ", "PATH_PREFIX": "", "CROP_SIZE": 224, "NUM_FRAMES": 8,
             "SAMPLING_RATE": 8, "MEAN": [0.45, 0.45, 0.45], "INPUT_CHANNEL_NUM": [3, 3], "STD": [0.225, 0.225, 0.225],
             "TRAIN_JITTER_SCALES": [256, 320], "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 256, "TARGET_FPS": 30,
             "DECODING_BACKEND": "pyav", "INV_UNIFORM_SAMPLE": False, "RANDOM_FLIP": True, "MULTI_LABEL": False,
             "ENSEMBLE_METHOD": "sum", "REVERSE_INPUT_CHANNEL": False, "FD": 0.0, "FIX_END": False,
             "TEMPORAL_EXTENT": 8, "DEIT_TRANSFORMS": False, "COLOR_JITTER": 0.0, "AUTO_AUGMENT": "", "RE_PROB": 0.0,
             "USE_RAND_AUGMENT": False, "USE_REPEATED_AUG": False, "USE_RANDOM_RESIZE_CROPS": False,
             "COLORJITTER": False, "GRAYSCALE": False, "GAUSSIAN": False},
    "SOLVER": {"BASE_LR": 0.1, "LR_POLICY": "cosine", "COSINE_END_LR": 0.0, "GAMMA": 0.1, "STEP_SIZE": 1, "STEPS": [],
               "LRS": [], "MAX_EPOCH": 300, "MOMENTUM": 0.9, "DAMPENING": 0.0, "NESTEROV": True, "WEIGHT_DECAY": 1e-4,
               "WARMUP_FACTOR": 0.1, "WARMUP_EPOCHS": 0.0, "WARMUP_START_LR": 0.01, "OPTIMIZING_METHOD": "sgd",
               "BASE_LR_SCALE_NUM_SHARDS": False},
    "NUM_GPUS": 1, "NUM_SHARDS": 1, "SHARD_ID": 0, "OUTPUT_DIR": "./tmp", "RNG_SEED": 1, "LOG_PERIOD": 10,
    "LOG_MODEL_INFO": False, "DIST_BACKEND": "nccl", "GLOBAL_BATCH_SIZE": 64,
    "BENCHMARK": {"NUM_EPOCHS": 5, "LOG_PERIOD": 100, "SHUFFLE": True},
    "DATA_LOADER": {"NUM_WORKERS": 8, "PIN_MEMORY": True, "ENABLE_MULTI_THREAD_DECODE": False},
    "DETECTION": {"ENABLE": False, "ALIGNED": True, "SPATIAL_SCALE_FACTOR": 16, "ROI_XFORM_RESOLUTION": 7},
    "EPICKITCHENS": {"VISUAL_DATA_DIR": "", "ANNOTATIONS_DIR": "", "TRAIN_LIST": "EPIC_100_train.pkl",
                     "VAL_LIST": "EPIC_100_validation.pkl", "TEST_LIST": "EPIC_100_validation.pkl",
                     "TEST_SPLIT": "validation", "TRAIN_PLUS_VAL": False, "ENABLE_ANTICIPATION": False},
    "MULTIGRID": {"EPOCH_FACTOR": 1.5, "SHORT_CYCLE": False, "LONG_CYCLE": False, "BN_BASE_SIZE": 8, "EVAL_FREQ": 3,
                  "LONG_CYCLE_SAMPLING_RATE": 0, "DEFAULT_B": 0, "DEFAULT_T": 0, "DEFAULT_S": 0},
    "TENSORBOARD": {"ENABLE": False, "LOG_DIR": ""},
    "SYNTHETIC": {"ENABLE": False, "NUM_VIDEOS": 16, "TEXT_LAYERS": 12},
}


def get_cfg():
    """A fresh copy of the default config (reference: lib/config/defaults.py:1073-1077)."""
    return CfgNode(copy.deepcopy(_DEFAULTS))
