"""Global mutable config, same access pattern as the reference's yacs ``cfg`` (core/cfgs.py:22-96).

yacs is not a dependency here: ``CfgNode`` is a minimal attribute-dict with merge_from_file / merge_from_list / dump.
Defaults = the reference defaults (core/cfgs.py:24-55) overlaid with configs/pymaf_config.yaml (the only experiment
config the reference ships), so ``cfg`` is usable without a yaml file.
"""
import ast
import copy


class CfgNode(dict):
    def __init__(self, init=None, new_allowed=False):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def _merge(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                if not isinstance(self.get(k), CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                if isinstance(v, str):
                    try:
                        v = ast.literal_eval(v)
                    except (ValueError, SyntaxError):
                        pass
                self[k] = v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split('.')
            for p in parts[:-1]:
                node = node[p]
            if isinstance(v, str):
                try:
                    v = ast.literal_eval(v)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = v

    def clone(self):
        return copy.deepcopy(self)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kw):
        import yaml
        return yaml.safe_dump(self.to_dict(), **kw)


cfg = CfgNode({
    'OUTPUT_DIR': 'results', 'DEVICE': 'cuda', 'DEBUG': False, 'LOGDIR': '', 'VAL_VIS_BATCH_FREQ': 200,
    'TRAIN_VIS_ITER_FERQ': 1000, 'SEED_VALUE': -1,
    'TRAIN': {'VAL_LOOP': False, 'STAGE': 2, 'NUM_WORKERS': 12, 'BATCH_SIZE': 64, 'PIN_MEMORY': False},
    'TEST': {'BATCH_SIZE': 32},
    'LOSS': {'KP_2D_W': 0., 'KP_3D_W': 300.0, 'SHAPE_W': 0.06, 'POSE_W': 60.0, 'VERT_W': 15.0, 'INDEX_WEIGHTS': 2.0,
             'PART_WEIGHTS': 0.3, 'POINT_REGRESSION_WEIGHTS': 0.125, 'FOCAL_WEIGHTS': 0.000001},
    'MODEL': {'PyMAF': {'MAF_ON': False, 'BACKBONE': 'vitpose', 'MLP_DIM': [256, 128, 64, 32], 'N_ITER': 3,
                        'AUX_SUPV_ON': True, 'DEPTH_SUPV_ON': False, 'FOCAL_SUPV_ON': False,
                        'DP_HEATMAP_SIZE': (128, 128)}},
    'RES_MODEL': {'DECONV_WITH_BIAS': False, 'NUM_DECONV_LAYERS': 3, 'NUM_DECONV_FILTERS': [256, 256, 256],
                  'NUM_DECONV_KERNELS': [4, 4, 4]},
    'IMG_RES': {'WIDTH': 256, 'HEIGHT': 256},
    'SOLVER': {'MAX_ITER': 500000, 'TYPE': 'Adam', 'BASE_LR': 0.00005, 'GAMMA': 0.1, 'STEPS': [0], 'EPOCHS': [0]},
})


def get_cfg_defaults():
    return cfg


def update_cfg(cfg_file):
    cfg.merge_from_file(cfg_file)
    return cfg


def parse_args(args):
    if getattr(args, 'cfg_file', None) is not None:
        update_cfg(args.cfg_file)
    if getattr(args, 'misc', None) is not None:
        cfg.merge_from_list(args.misc)
    return cfg
