"""Stub of `omegaconf` (absent offline): only the names src/global_cfg.py:3 imports."""


class DictConfig(dict):
    pass


class OmegaConf:
    pass
