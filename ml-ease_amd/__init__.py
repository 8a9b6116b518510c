"""ml-ease_amd: MI355X-native drop-in for the ADMM L2-logistic hot path of linkedin/ml-ease.

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C-ABI of include/mlease_admm.h),
the host-side mirror of the reference's driver loop (``admm.py``), row preparation / partition
indexing (``dataset.py``) and the avro container formats (``avro_io.py``).

The directory name carries a hyphen (repo contract); import it as ``mlease_amd`` through the
loader shim ``mlease_amd.py`` at the repository root.
"""
from . import admm, avro_io, dataset, hip_engine  # noqa: F401

__all__ = ["admm", "avro_io", "dataset", "hip_engine"]
