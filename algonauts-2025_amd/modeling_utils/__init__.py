"""modeling_utils (MI355X build): the encoder-op plugin surface of the reference package
(`models`, `losses`, `metrics`) with the arithmetic executed by hand-written gfx950 HIP
kernels through libtribe_hip.so.  Same class / config names and call signatures as
/root/reference/modeling_utils/modeling_utils."""
from . import losses, metrics, models  # noqa: F401
