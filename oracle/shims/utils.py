"""Oracle shim (TEST INFRASTRUCTURE ONLY): replaces the reference's top-level `utils.py`
for import purposes.  models/modeling_distributed_gpt3.py:36 does `from utils import File`
and uses it only in save_checkpoint (:1518), which the oracle never calls; the real
utils.py cannot be imported here (needs `sh`, `timm.utils`, tensorboard)."""


class File:
    @staticmethod
    def write(obj, path):  # pragma: no cover - never used by the oracle
        with open(path, "wb") as f:
            f.write(obj)
